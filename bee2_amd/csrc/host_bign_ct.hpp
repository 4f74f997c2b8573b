// host_bign_ct.hpp -- the drop-in layer's CONSTANT-TIME host path for ONE public-key calculation / key generation /
// signature on a standard bign curve (product code; plain C++17, no HIP).
//
// Why it exists (VERDICT r03 "missing 3", item 5): one bign128Sign2 through the GPU is three kernels of a lone wavefront plus
// the copies -- 187 us -- where the reference signs in 39 us on a host core; a bee2 program relinked against libbee2hip.so
// that signs ONE message at a time got five times slower.  The reference itself signs on the host, with regular table
// multipliers (bignMulBase, src/crypto/bign/bign_misc.c:115-137 -> ecMulAPre*, src/math/ec.c:626-1121); this header does the
// same job its own way and at least as carefully:
//   * Q = d G / R = k G: fixed-base comb over SIGNED 6-bit windows (43 / 65 / 86 windows, no doublings), the formulation of the
//     GPU kernel bign_mulbase_ct_kernel (bign_sign_kernels.hip mul_base_ct6<N, true>, checked by tools/model_sign_w6.py): every
//     window scans ALL 32 entries of its table row and keeps the wanted one with masks (no secret index ever addresses
//     memory), negates y by mask, and adds with the Jacobian mixed formula 8M + 3S whose exceptional cases the window
//     schedule excludes (see there); "accumulator still at infinity" and "digit is zero" are masks, not branches;
//   * GF(p): 64-bit limbs, Crandall folds, every correction a masked add / subtract of c -- no early exit, no data-dependent
//     branch (the verification side's Field<N> in host_bign.hpp branches freely; nothing secret goes there);
//   * 1 / Z by a^(p - 2): the exponent is public, the chain fixed;
//   * mod q: schoolbook product, four folds by 2^(2l) - q, one masked subtraction; the two zzSubMod steps of
//     bign_sign.c:232-236 bit for bit (H is NOT reduced first);
//   * every secret temporary (d, k, theta, digits, selected points, accumulators) is wiped before return.
// What is NOT constant-time here, exactly as in the reference: belt.  theta = belt-hash(oid || d || t) and k = belt-wbl_theta(H)
// run on hostp::belt_encr, a table-driven belt like bee2's own beltBlockEncr (src/crypto/belt/belt_block.c:210-269, H5 / H13 /
// H21 / H29 lookups at secret-dependent indices): same cache-timing surface as the library this one replaces.  A caller who
// wants the private key to meet no data cache at all sets BEE2HIP_FORCE=gpu (bee2hip_path_policy(1)): then every bign
// operation with a secret runs in the GPU kernels, whose S-box copies are bank-private (bign_sign_kernels.hip).
//
// Used ONLY by the drop-in symbols bignPubkeyCalc / bignKeypairGen / bignSign / bignSign2 and their bign128 / 192 / 256
// facades for ONE item on a standard parameter set, in auto / cpu mode, after the calling thread has initialised its HIP
// device.  Batch / _dev / _multi entry points never come here.  tests/test_host_bign_ct.py pins it on CPU to the
// reference's fixtures (tests/golden/bign_sign.json) and to the oracle; tools/ct_audit_x86.py lists the conditional
// branches of the compiled object; tests/test_gpu_bign_sign.py runs every signing fixture in both modes.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "host_bign.hpp"

namespace bee2hip {
namespace hostct {

using hostb::u128;
using hostb::Fe;
using hostb::Aff;
constexpr uint32_t kOk = 0, kBadRng = 304, kBadParams = 502, kBadPrivkey = 504;      // include/bee2/core/err.h

// ---- masks -------------------------------------------------------------------------------------------------------
// The optimiser must not know that a mask is 0 or all-ones: clang -O3 turns "x + (c & -carry)" into "if (carry) x += c" -- a
// branch on secret data (found by tools/ct_audit_x86.py: `jae` after the adds of reduce / add / sub).  An empty asm that
// claims to modify the value hides its origin; every mask is made through it.
static inline uint64_t m_hide(uint64_t x)
{
#if defined(__GNUC__) || defined(__clang__)
    __asm__("" : "+r"(x));
#endif
    return x;
}
static inline uint64_t m_bit(uint64_t bit01) { return m_hide((uint64_t)0 - bit01); }            // 0 / 1 -> 0 / all-ones
static inline uint64_t m_zero(uint64_t x) { return m_bit((~x & (x - 1)) >> 63); }               // all-ones iff x == 0
static inline uint64_t m_eq(uint64_t a, uint64_t b) { return m_zero(a ^ b); }
static inline uint64_t m_sel(uint64_t m, uint64_t a, uint64_t b) { return b ^ (m & (a ^ b)); }  // m ? a : b
// a wipe the optimiser may not drop
static inline void wipe(void *p, size_t n)
{
    // every buffer wiped here is an array of 64-bit (or 32-bit) words: whole words, then whatever is left
    volatile uint64_t *w = (volatile uint64_t *)p;
    for (; n >= 8; n -= 8) *w++ = 0;
    volatile unsigned char *q = (volatile unsigned char *)w;
    while (n--) *q++ = 0;
}
template <int N> static inline uint64_t m_is_zero(const uint64_t (&a)[N])
{
    uint64_t acc = 0;
    for (int i = 0; i < N; ++i) acc |= a[i];
    return m_zero(acc);
}
// all-ones iff a < b
template <int N> static inline uint64_t m_lt(const uint64_t (&a)[N], const uint64_t (&b)[N])
{
    uint64_t borrow = 0;
    for (int i = 0; i < N; ++i) {
        const u128 d = (u128)a[i] - b[i] - borrow;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return m_bit(borrow);
}

// ---- GF(p), p = 2^(64 N) - c.  Values are kept WEAKLY reduced: any N-limb number (the class mod p); canon() at the end.
template <int N>
struct FieldCt {
    uint64_t c;
    static void sel(Fe<N> &r, uint64_t m, const Fe<N> &a, const Fe<N> &b)
    {
        for (int i = 0; i < N; ++i) r.v[i] = m_sel(m, a.v[i], b.v[i]);
    }
    void add(Fe<N> &r, const Fe<N> &a, const Fe<N> &b) const
    {
        uint64_t t[N], carry = 0;
        for (int i = 0; i < N; ++i) { const u128 s = (u128)a.v[i] + b.v[i] + carry; t[i] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
        // 2^(64 N) = c (mod p): the carry comes back as + c; that can wrap once more (then what is left is below c)
        for (int pass = 0; pass < 2; ++pass) {
            u128 s = (u128)t[0] + (c & m_bit(carry));
            t[0] = (uint64_t)s;
            uint64_t k = (uint64_t)(s >> 64);
            for (int i = 1; i < N; ++i) { s = (u128)t[i] + k; t[i] = (uint64_t)s; k = (uint64_t)(s >> 64); }
            carry = k;
        }
        for (int i = 0; i < N; ++i) r.v[i] = t[i];
    }
    void sub(Fe<N> &r, const Fe<N> &a, const Fe<N> &b) const
    {
        uint64_t t[N], borrow = 0;
        for (int i = 0; i < N; ++i) { const u128 d = (u128)a.v[i] - b.v[i] - borrow; t[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
        for (int pass = 0; pass < 2; ++pass) {       // - 2^(64 N) = - c (mod p)
            u128 d = (u128)t[0] - (c & m_bit(borrow));
            t[0] = (uint64_t)d;
            uint64_t k = (uint64_t)(d >> 64) & 1;
            for (int i = 1; i < N; ++i) { d = (u128)t[i] - k; t[i] = (uint64_t)d; k = (uint64_t)(d >> 64) & 1; }
            borrow = k;
        }
        for (int i = 0; i < N; ++i) r.v[i] = t[i];
    }
    void neg(Fe<N> &r, const Fe<N> &a) const
    {
        Fe<N> z;
        memset(&z, 0, sizeof z);
        sub(r, z, a);
    }
    void dbl(Fe<N> &r, const Fe<N> &a) const { add(r, a, a); }
    void reduce(Fe<N> &r, const uint64_t t[2 * N]) const
    {
        uint64_t carry = 0;
        for (int i = 0; i < N; ++i) {
            const u128 m = (u128)t[N + i] * c + t[i] + carry;
            r.v[i] = (uint64_t)m;
            carry = (uint64_t)(m >> 64);
        }
        u128 m = (u128)carry * c + r.v[0];             // carry <= c
        r.v[0] = (uint64_t)m;
        uint64_t k = (uint64_t)(m >> 64);
        for (int i = 1; i < N; ++i) { const u128 s = (u128)r.v[i] + k; r.v[i] = (uint64_t)s; k = (uint64_t)(s >> 64); }
        m = (u128)r.v[0] + (c & m_bit(k));               // wrapped once more: what is left is below c^2, + c cannot wrap
        r.v[0] = (uint64_t)m;
        k = (uint64_t)(m >> 64);
        for (int i = 1; i < N; ++i) { const u128 s = (u128)r.v[i] + k; r.v[i] = (uint64_t)s; k = (uint64_t)(s >> 64); }
    }
    void mul(Fe<N> &r, const Fe<N> &a, const Fe<N> &b) const
    {
        uint64_t t[2 * N];
        for (int j = 0; j < N; ++j) t[j] = 0;
        for (int i = 0; i < N; ++i) {
            uint64_t carry = 0;
            for (int j = 0; j < N; ++j) {
                const u128 m = (u128)a.v[i] * b.v[j] + t[i + j] + carry;
                t[i + j] = (uint64_t)m;
                carry = (uint64_t)(m >> 64);
            }
            t[i + N] = carry;
        }
        reduce(r, t);
        wipe(t, sizeof t);
    }
    // a^2: the products a_i a_j, i < j, once and doubled, plus the squares (N (N + 1) / 2 multiplications instead of N^2)
    void sqr(Fe<N> &r, const Fe<N> &a) const
    {
        uint64_t t[2 * N];
        for (int j = 0; j < 2 * N; ++j) t[j] = 0;
        for (int i = 0; i + 1 < N; ++i) {
            uint64_t carry = 0;
            for (int j = i + 1; j < N; ++j) {
                const u128 m = (u128)a.v[i] * a.v[j] + t[i + j] + carry;
                t[i + j] = (uint64_t)m;
                carry = (uint64_t)(m >> 64);
            }
            t[i + N] = carry;
        }
        uint64_t top = 0;
        for (int k = 0; k < 2 * N; ++k) { const uint64_t nt = t[k] >> 63; t[k] = (t[k] << 1) | top; top = nt; }
        uint64_t carry = 0;
        for (int i = 0; i < N; ++i) {
            const u128 m = (u128)a.v[i] * a.v[i] + t[2 * i] + carry;
            t[2 * i] = (uint64_t)m;
            const u128 s = (u128)t[2 * i + 1] + (uint64_t)(m >> 64);
            t[2 * i + 1] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        reduce(r, t);
        wipe(t, sizeof t);
    }
    // the representative in [0, p): a >= p  <=>  a + c carries out, and then a - p = a + c - 2^(64 N)
    void canon(Fe<N> &r, const Fe<N> &a) const
    {
        uint64_t s[N];
        u128 w = (u128)a.v[0] + c;
        s[0] = (uint64_t)w;
        for (int i = 1; i < N; ++i) { w = (u128)a.v[i] + (uint64_t)(w >> 64); s[i] = (uint64_t)w; }
        const uint64_t ge = m_bit((uint64_t)(w >> 64));
        for (int i = 0; i < N; ++i) r.v[i] = m_sel(ge, s[i], a.v[i]);
    }
    // a^(p - 2): the exponent is public (2^(64 N) - c - 2), so its windows may steer the code; 0 -> 0
    void inv(Fe<N> &r, const Fe<N> &a) const
    {
        Fe<N> tab[16];
        memset(&tab[0], 0, sizeof tab[0]);
        tab[0].v[0] = 1;
        tab[1] = a;
        for (int i = 2; i < 16; ++i) mul(tab[i], tab[i - 1], a);
        uint64_t e[N];
        for (int i = 0; i < N; ++i) e[i] = ~(uint64_t)0;
        e[0] = (uint64_t)0 - (c + 2);
        Fe<N> x = tab[15];
        for (int bit = 64 * N - 8; bit >= 0; bit -= 4) {
            for (int k = 0; k < 4; ++k) sqr(x, x);
            const int w = (int)(e[bit >> 6] >> (bit & 63)) & 15;       // public
            if (w) mul(x, x, tab[w]);
        }
        r = x;
        wipe(tab, sizeof tab);
        wipe(&x, sizeof x);
    }
};

template <int N> struct JacCt { Fe<N> X, Y, Z; };
template <int N> struct Win6 { static constexpr int W = (64 * N + 1 + 5) / 6; };     // 43 / 65 / 86

// T <- T + E, T Jacobian (not the point at infinity), E affine, T != +-E: the 8M + 3S formula of the GPU kernel
// (bign_sign_kernels.hip jac_madd_ct).  The callers' window schedule keeps the exceptional cases away; a result computed
// from a dummy operand is discarded by mask.
template <int N>
static inline void jac_madd(const FieldCt<N> &F, JacCt<N> &T, const Aff<N> &E)
{
    Fe<N> Z1Z1, U2, S2, H, HH, HHH, r, V, t;
    F.sqr(Z1Z1, T.Z);
    F.mul(U2, E.x, Z1Z1);
    F.mul(t, T.Z, Z1Z1);
    F.mul(S2, E.y, t);
    F.sub(H, U2, T.X);
    F.sub(r, S2, T.Y);
    F.sqr(HH, H);
    F.mul(HHH, H, HH);
    F.mul(V, T.X, HH);
    F.mul(T.Z, T.Z, H);
    F.sqr(t, r);
    F.sub(t, t, HHH);
    F.dbl(U2, V);
    F.sub(T.X, t, U2);
    F.sub(t, V, T.X);
    F.mul(t, r, t);
    F.mul(S2, T.Y, HHH);
    F.sub(T.Y, t, S2);
    wipe(&Z1Z1, sizeof Z1Z1); wipe(&U2, sizeof U2); wipe(&S2, sizeof S2); wipe(&H, sizeof H); wipe(&HH, sizeof HH);
    wipe(&HHH, sizeof HHH); wipe(&r, sizeof r); wipe(&V, sizeof V); wipe(&t, sizeof t);
}

// One standard curve with what the signing side needs: q, 2^(64 N) - q, b is not needed (a = -3 formulas), and the table
// tab6[w * 32 + j - 1] = j 2^(6 w) G, affine, canonical, j = 1 .. 32 -- PUBLIC data, built once with the variable-time
// group law of host_bign.hpp.
template <int N>
struct SignCurve {
    FieldCt<N> F;
    uint64_t q[N], cq[N];
    Aff<N> *tab6 = nullptr;
    bool ready = false;

    void init(const hostb::Curve<N> &E)
    {
        constexpr int W = Win6<N>::W;
        F.c = E.F.c;
        uint64_t borrow = 0;
        for (int i = 0; i < N; ++i) {
            q[i] = E.q[i];
            const u128 d = (u128)0 - E.q[i] - borrow;          // 2^(64 N) - q
            cq[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        tab6 = (Aff<N> *)malloc(sizeof(Aff<N>) * (size_t)W * 32);
        hostb::Jac<N> *J = (hostb::Jac<N> *)malloc(sizeof(hostb::Jac<N>) * (size_t)W * 32);
        Fe<N> *pre = (Fe<N> *)malloc(sizeof(Fe<N>) * (size_t)W * 32);
        if (!tab6 || !J || !pre) { free(tab6); free(J); free(pre); tab6 = nullptr; return; }
        hostb::Jac<N> B;
        E.from_aff(B, E.tabG[0][0], false);                    // G
        for (int w = 0; w < W; ++w) {
            J[w * 32] = B;
            for (int j = 1; j < 32; ++j) E.add(J[w * 32 + j], J[w * 32 + j - 1], B, false);
            for (int k = 0; k < 6; ++k) E.dbl(B, B);
        }
        // all to affine with one inversion; a multiple of G that happens to be O (j 2^(6w) = 0 mod q: not for these sizes)
        // would have Z = 0 and poison the product -- checked
        const int n = W * 32;
        bool ok = true;
        pre[0] = J[0].Z;
        for (int i = 1; i < n; ++i) { ok = ok && !hostb::Field<N>::is_zero(J[i].Z); E.F.mul(pre[i], pre[i - 1], J[i].Z); }
        Fe<N> acc, zi, zi2;
        E.F.inv(acc, pre[n - 1]);
        for (int i = n - 1; i >= 0; --i) {
            if (i) { E.F.mul(zi, acc, pre[i - 1]); E.F.mul(acc, acc, J[i].Z); } else zi = acc;
            E.F.sqr(zi2, zi);
            E.F.mul(tab6[i].x, J[i].X, zi2);
            E.F.mul(zi2, zi2, zi);
            E.F.mul(tab6[i].y, J[i].Y, zi2);
        }
        free(J);
        free(pre);
        ready = ok;
    }

    uint64_t in_range_q(const uint64_t (&x)[N]) const { return ~m_is_zero<N>(x) & m_lt<N>(x, q); }     // all-ones iff 0 < x < q

    // (x, y) = k G, canonical; returns all-ones iff k G = O (k = 0 mod q), then x = y = 0.  k: any N-limb value.
    uint64_t mul_base(Fe<N> &x, Fe<N> &y, const uint64_t (&k)[N]) const
    {
        constexpr int W = Win6<N>::W;
        uint64_t kk[N];
        for (int i = 0; i < N; ++i) kk[i] = k[i];
        JacCt<N> J, sum;
        Fe<N> one, ny;
        Aff<N> E;
        memset(&J, 0, sizeof J);
        memset(&one, 0, sizeof one);
        one.v[0] = 1;
        J.Y = one;
        uint64_t at_inf = ~(uint64_t)0, carry = 0;
        for (int w = 0; w < W; ++w) {
            const uint64_t t = (kk[0] & 63u) + carry;                       // 0 .. 64
            for (int i = 0; i + 1 < N; ++i) kk[i] = (kk[i] >> 6) | (kk[i + 1] << 58);
            kk[N - 1] >>= 6;
            carry = (t + 32u) >> 6;                                         // 1 iff t >= 32
            const uint64_t d = t - (carry << 6);                            // the digit, two's complement, -32 .. 31
            const uint64_t neg = m_hide((uint64_t)((int64_t)d >> 63));
            const uint64_t mag = (d ^ neg) - neg;                           // 0 .. 32
            // the whole row, every time: the addresses depend on w only
            const Aff<N> *row = tab6 + (size_t)w * 32;
            memset(&E, 0, sizeof E);
            for (uint64_t j = 1; j <= 32; ++j) {
                const uint64_t m = m_eq(mag, j);
                for (int l = 0; l < N; ++l) { E.x.v[l] |= row[j - 1].x.v[l] & m; E.y.v[l] |= row[j - 1].y.v[l] & m; }
            }
            F.neg(ny, E.y);
            FieldCt<N>::sel(E.y, neg, ny, E.y);
            const uint64_t keep = m_zero(mag);                              // digit 0: the accumulator stays
            sum = J;
            jac_madd(F, sum, E);                                            // digit 0 or accumulator still O: computed, not used
            const uint64_t set = at_inf & ~keep;                            // first non-zero digit: J <- (x, y, 1)
            for (int l = 0; l < N; ++l) {
                J.X.v[l] = m_sel(keep, J.X.v[l], m_sel(set, E.x.v[l], sum.X.v[l]));
                J.Y.v[l] = m_sel(keep, J.Y.v[l], m_sel(set, E.y.v[l], sum.Y.v[l]));
                J.Z.v[l] = m_sel(keep, J.Z.v[l], m_sel(set, one.v[l], sum.Z.v[l]));
            }
            at_inf &= keep;
        }
        for (int l = 0; l < N; ++l) J.Z.v[l] &= ~at_inf;                    // nothing ever added: k = 0
        Fe<N> zc, zi, zi2;
        F.canon(zc, J.Z);
        const uint64_t inf = m_is_zero<N>(zc.v);
        F.inv(zi, zc);
        F.sqr(zi2, zi);
        F.mul(x, J.X, zi2);
        F.mul(zi2, zi2, zi);
        F.mul(y, J.Y, zi2);
        F.canon(x, x);
        F.canon(y, y);
        for (int l = 0; l < N; ++l) { x.v[l] &= ~inf; y.v[l] &= ~inf; }
        wipe(kk, sizeof kk); wipe(&J, sizeof J); wipe(&sum, sizeof sum); wipe(&E, sizeof E); wipe(&ny, sizeof ny);
        wipe(&zc, sizeof zc); wipe(&zi, sizeof zi); wipe(&zi2, sizeof zi2);
        return inf;
    }

    // x (2 N limbs) mod q -> [0, q): 2^(64 N) = cq (mod q), cq < 2^(32 N + 32); four folds bring any 2N-limb value below
    // 2^(64 N) (< 2^(96N+32), < 2^(64N + 66), <= 2^(64 N) + 2^(32N + 98), then no carry), one masked subtraction finishes
    void mod_q(uint64_t (&r)[N], const uint64_t (&x)[2 * N]) const
    {
        uint64_t a[2 * N], t[2 * N];
        for (int i = 0; i < 2 * N; ++i) a[i] = x[i];
        for (int pass = 0; pass < 4; ++pass) {
            for (int i = 0; i < 2 * N; ++i) t[i] = 0;
            for (int i = 0; i < N; ++i) {               // hi * cq
                uint64_t carry = 0;
                for (int j = 0; j < N; ++j) {
                    const u128 m = (u128)a[N + i] * cq[j] + t[i + j] + carry;
                    t[i + j] = (uint64_t)m;
                    carry = (uint64_t)(m >> 64);
                }
                t[i + N] = carry;
            }
            uint64_t carry = 0;                         // + lo
            for (int i = 0; i < 2 * N; ++i) {
                const u128 s = (u128)t[i] + (i < N ? a[i] : 0) + carry;
                a[i] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
        }
        uint64_t s[N], borrow = 0;
        for (int i = 0; i < N; ++i) { const u128 d = (u128)a[i] - q[i] - borrow; s[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
        const uint64_t lt = m_bit(borrow);              // a < q: keep a
        for (int i = 0; i < N; ++i) r[i] = m_sel(lt, a[i], s[i]);
        wipe(a, sizeof a); wipe(t, sizeof t); wipe(s, sizeof s);
    }
    // zzSubMod (src/math/zz/zz_mod.c:120-132): c = a - b, + q when the subtraction borrowed; b need not be below q
    void sub_mod_q(uint64_t (&c)[N], const uint64_t (&a)[N], const uint64_t (&b)[N]) const
    {
        uint64_t t[N], borrow = 0;
        for (int i = 0; i < N; ++i) { const u128 d = (u128)a[i] - b[i] - borrow; t[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
        const uint64_t m = m_bit(borrow);
        uint64_t carry = 0;
        for (int i = 0; i < N; ++i) { const u128 s = (u128)t[i] + (q[i] & m) + carry; c[i] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
        wipe(t, sizeof t);
    }
};

template <int N> static inline void ld_limbs(uint64_t (&r)[N], const uint8_t *p) { for (int i = 0; i < N; ++i) r[i] = hostp::ld64le(p + 8 * i); }
template <int N> static inline void st_limbs(uint8_t *p, const uint64_t (&a)[N]) { for (int i = 0; i < N; ++i) hostp::st64le(p + 8 * i, a[i]); }

// bignPubkeyCalc (bign_misc.c:373-417): Q = d G for 0 < d < q, else ERR_BAD_PRIVKEY and nothing written.
// keygen (bignKeypairGen, bign_misc.c:182-229, after the rng): any d is multiplied; ERR_BAD_PARAMS when d G = O.
template <int N>
static inline uint32_t pubkey_calc(const SignCurve<N> &C, bool keygen, const uint8_t *privkey, uint8_t *pubkey)
{
    uint64_t d[N];
    ld_limbs<N>(d, privkey);
    const uint64_t ok = keygen ? ~(uint64_t)0 : C.in_range_q(d);
    Fe<N> x, y;
    const uint64_t inf = C.mul_base(x, y, d);
    wipe(d, sizeof d);
    // the verdicts are public (the caller sees the error code)
    uint32_t code = kOk;
    if (!ok) code = kBadPrivkey;
    else if (keygen && inf) code = kBadParams;
    if (code == kOk) { st_limbs<N>(pubkey, x.v); st_limbs<N>(pubkey + 8 * N, y.v); }
    wipe(&x, sizeof x); wipe(&y, sizeof y);
    return code;
}

// bignSign2 (bign_sign.c:140-245; k_in == null: k by algorithm 6.3.3 from theta = belt-hash(oid || d || t)) and bignSign after
// its generator (bign_sign.c:32-112; k_in = the one-time key drawn by the caller's rng).  sig = s0 (4 N octets) || s1 (8 N).
template <int N>
static inline uint32_t sign(const SignCurve<N> &C, const hostp::BeltTables &T, const uint8_t Hbox[256], const uint8_t *oid_der,
                            size_t oid_len, const uint8_t *hash, const uint8_t *privkey, const uint8_t *k_in, const void *t,
                            size_t t_len, uint8_t *sig)
{
    constexpr int no = 8 * N, NB = no / 16, HN = N / 2;
    uint64_t d[N], k[N], Hh[N];
    ld_limbs<N>(d, privkey);
    ld_limbs<N>(Hh, hash);
    if (!C.in_range_q(d)) { wipe(d, sizeof d); return kBadPrivkey; }          // public verdict (:185-189, :62-68)
    if (k_in) {
        ld_limbs<N>(k, k_in);
        if (!C.in_range_q(k)) { wipe(d, sizeof d); wipe(k, sizeof k); return kBadRng; }
    } else {
        // theta = belt-hash(oid || d || t) (:192-199)
        uint8_t theta[32];
        {
            hostb::BeltHashPieces bh(T, Hbox);
            bh.absorb(oid_der, oid_len);
            bh.absorb(privkey, no);
            if (t && t_len) bh.absorb((const uint8_t *)t, t_len);
            bh.digest(theta);
            wipe(&bh.hs, sizeof bh.hs); wipe(bh.block, sizeof bh.block);
        }
        uint32_t key[8];
        for (int i = 0; i < 8; ++i) key[i] = hostp::ld32le(theta + 4 * i);
        // k <- H; repeat k <- belt-wbl_theta(k) until 0 < k < q (:201-216).  belt-wbl on NB blocks (belt_wbl.c:58-152):
        // 2 NB rounds  s = r_1 ^ .. ^ r_{NB-1};  (r_1 .. r_NB) <- (r_2, .., r_{NB-1}, r_NB ^ E(s) ^ <i>, s)
        uint32_t r[NB][4];
        for (int j = 0; j < NB; ++j) for (int i = 0; i < 4; ++i) r[j][i] = hostp::ld32le(hash + 16 * j + 4 * i);
        for (;;) {                                   // leaves with probability > 1 - 2^-126 per pass on the standard curves
            for (int round = 1; round <= 2 * NB; ++round) {
                uint32_t s[4], e[4], last[4];
                for (int i = 0; i < 4; ++i) {
                    s[i] = r[0][i];
                    for (int j = 1; j < NB - 1; ++j) s[i] ^= r[j][i];
                    e[i] = s[i];
                }
                hostp::belt_encr(T, e, key);
                e[0] ^= (uint32_t)round;
                for (int i = 0; i < 4; ++i) last[i] = r[NB - 1][i] ^ e[i];
                for (int j = 0; j + 2 < NB; ++j) for (int i = 0; i < 4; ++i) r[j][i] = r[j + 1][i];
                for (int i = 0; i < 4; ++i) { r[NB - 2][i] = last[i]; r[NB - 1][i] = s[i]; }
                wipe(s, sizeof s); wipe(e, sizeof e); wipe(last, sizeof last);
            }
            for (int i = 0; i < N; ++i) k[i] = (uint64_t)r[i / 2][(2 * i) & 3] | (uint64_t)r[i / 2][((2 * i) & 3) + 1] << 32;
            if (C.in_range_q(k)) break;              // as the reference: the exit is the one data-dependent branch (:206-216)
        }
        wipe(theta, sizeof theta); wipe(key, sizeof key); wipe(r, sizeof r);
    }
    // R = k G (:218-224); k != 0 mod q, so R != O
    Fe<N> x, y;
    (void)C.mul_base(x, y, k);
    uint8_t rx[no], hs[32];
    st_limbs<N>(rx, x.v);
    // s0 = belt-hash(oid || <x_R> || H) mod 2^l (:226-231): public values
    {
        hostb::BeltHashPieces bh(T, Hbox);
        bh.absorb(oid_der, oid_len);
        bh.absorb(rx, no);
        bh.absorb(hash, no);
        bh.digest(hs);
    }
    // s1 = (k - (s0 + 2^l) d - H) mod q (:232-238)
    uint64_t s0[HN + 1], prod[2 * N], s1[N];
    for (int i = 0; i < HN; ++i) s0[i] = hostp::ld64le(hs + 8 * i);
    s0[HN] = 1;
    for (int i = 0; i < 2 * N; ++i) prod[i] = 0;
    for (int i = 0; i <= HN; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < N; ++j) {
            const u128 m = (u128)s0[i] * d[j] + prod[i + j] + carry;
            prod[i + j] = (uint64_t)m;
            carry = (uint64_t)(m >> 64);
        }
        prod[i + N] = carry;
    }
    C.mod_q(s1, prod);
    C.sub_mod_q(s1, k, s1);
    C.sub_mod_q(s1, s1, Hh);
    memcpy(sig, hs, no / 2);
    st_limbs<N>(sig + no / 2, s1);
    wipe(d, sizeof d); wipe(k, sizeof k); wipe(prod, sizeof prod); wipe(s1, sizeof s1); wipe(&x, sizeof x); wipe(&y, sizeof y);
    return kOk;
}

}  // namespace hostct
}  // namespace bee2hip
