// host_bign.hpp -- the drop-in layer's HOST path for ONE bign signature verification (product code; plain C++17, no HIP).
//
// Why it exists: one bign128Verify through the GPU is a 0.41-0.45 ms call (upload, five kernels of one busy lane-quad,
// download; profiles/r03_single_call_kernels.txt) where the reference needs 0.18 ms on a host core -- VERDICT r02
// "missing 3": a bee2 program relinked against libbee2hip.so that verifies ONE signature (cmd/core/cmd_sig.c:484-490 on a
// single file) got slower, not faster.  This header verifies a single signature on the calling core in 27 us on the
// 256-bit curve (and validates ONE public key in a fraction of a microsecond: two squarings and a product).  Same rules
// as host_small.hpp: used ONLY by the drop-in symbols bignVerify / bign128Verify / bign192Verify / bign256Verify /
// bignPubkeyVal / bign128PubkeyVal / ... on one of the three standard parameter sets (capi.hip), never by a bee2hip_*_batch /
// *_dev / *_multi entry point or a timed region of bench.py; only after the calling thread has initialised its HIP
// device; BEE2HIP_FORCE=gpu keeps every call on the GPU.  Verification handles no secrets: nothing here is constant-time,
// and nothing on the signing side (private keys, one-time keys) may be routed through this file.
//
// It is an independent statement of STB 34.101.45 7.1.4 as the reference runs it (bignVerifyEc,
// src/crypto/bign/bign_sign.c:268-344) -- not the oracle, not the reference's code: 64-bit limbs with Crandall
// reduction (what zmMulCrand / zzRedCrand compute, src/math/zm.c:214-253, zz_red.c:71-105), Jacobian coordinates with
// a = -3 (the curves of bign_params.c:34-140), and -- where the reference interleaves two width-5 NAFs over 2l
// doublings (ecAddMulA, src/math/ec.c:1183-1273) -- THREE interleaved NAFs over l + 1 doublings: u = u0 + 2^l u1 against
// fixed affine tables of G and 2^l G (odd multiples 1..63, built once per curve), v = s0 + 2^l against the odd multiples
// 1..15 of Q.  Every exceptional case of the group law is handled explicitly (variable time is fine here), so R = O is
// detected exactly and the verdicts are the reference's (tests/test_host_bign.py: the 433 edge-case fixtures, G.2 / G.3,
// all three curves; tests/test_gpu_hostpath.py runs the drop-in fixtures through both paths on the GPU box).
#pragma once
#include <stdint.h>
#include <string.h>

#include "host_small.hpp"

namespace bee2hip {
namespace hostb {

typedef unsigned __int128 u128;
constexpr uint32_t kOk = 0, kBadPubkey = 505, kBadSig = 510;      // include/bee2/core/err.h:72,186,196

template <int N> struct Fe { uint64_t v[N]; };

// GF(p), p = 2^(64 N) - c, elements canonical in [0, p)
template <int N>
struct Field {
    uint64_t c;
    static bool is_zero(const Fe<N> &a)
    {
        uint64_t acc = 0;
        for (int i = 0; i < N; ++i) acc |= a.v[i];
        return acc == 0;
    }
    static bool eq(const Fe<N> &a, const Fe<N> &b)
    {
        uint64_t acc = 0;
        for (int i = 0; i < N; ++i) acc |= a.v[i] ^ b.v[i];
        return acc == 0;
    }
    // a >= p  <=>  a + c carries out of 2^(64 N)
    bool ge_p(const Fe<N> &a) const
    {
        u128 s = (u128)a.v[0] + c;
        for (int i = 1; i < N; ++i) s = (u128)a.v[i] + (uint64_t)(s >> 64);
        return (uint64_t)(s >> 64) != 0;
    }
    void add_c(uint64_t w[N]) const        // w <- (w + c) mod 2^(64 N)
    {
        u128 s = (u128)w[0] + c;
        w[0] = (uint64_t)s;
        for (int i = 1; i < N && (uint64_t)(s >> 64); ++i) { s = (u128)w[i] + 1; w[i] = (uint64_t)s; }
    }
    void add(Fe<N> &r, const Fe<N> &a, const Fe<N> &b) const
    {
        uint64_t carry = 0;
        Fe<N> t;
        for (int i = 0; i < N; ++i) { const u128 s = (u128)a.v[i] + b.v[i] + carry; t.v[i] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
        if (carry || ge_p(t)) add_c(t.v);              // a + b - p = a + b + c - 2^(64 N)
        r = t;
    }
    void sub(Fe<N> &r, const Fe<N> &a, const Fe<N> &b) const
    {
        uint64_t borrow = 0;
        Fe<N> t;
        for (int i = 0; i < N; ++i) {
            const u128 d = (u128)a.v[i] - b.v[i] - borrow;
            t.v[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        if (borrow) {                                  // + p = - c (mod 2^(64 N)); cannot borrow again: t >= 2^(64 N) - p + 1
            u128 d = (u128)t.v[0] - c;
            t.v[0] = (uint64_t)d;
            for (int i = 1; i < N && ((uint64_t)(d >> 64) & 1); ++i) { d = (u128)t.v[i] - 1; t.v[i] = (uint64_t)d; }
        }
        r = t;
    }
    void neg(Fe<N> &r, const Fe<N> &a) const
    {
        Fe<N> z;
        memset(&z, 0, sizeof z);
        sub(r, z, a);
    }
    void dbl(Fe<N> &r, const Fe<N> &a) const { add(r, a, a); }
    // t[2N] -> r: two folds of the upper half times c, then one conditional subtraction (zz_red.c:40-48,71-105)
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
        for (int i = 1; i < N && k; ++i) { const u128 s = (u128)r.v[i] + k; r.v[i] = (uint64_t)s; k = (uint64_t)(s >> 64); }
        if (k) add_c(r.v);                             // wrapped once more: what is left is below c^2, + c cannot wrap
        if (ge_p(r)) add_c(r.v);
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
    }
    void sqr(Fe<N> &r, const Fe<N> &a) const
    {
        uint64_t t[2 * N];
        for (int j = 0; j < 2 * N; ++j) t[j] = 0;
        for (int i = 0; i + 1 < N; ++i) {              // the products a_i a_j, i < j, once
            uint64_t carry = 0;
            for (int j = i + 1; j < N; ++j) {
                const u128 m = (u128)a.v[i] * a.v[j] + t[i + j] + carry;
                t[i + j] = (uint64_t)m;
                carry = (uint64_t)(m >> 64);
            }
            t[i + N] = carry;
        }
        uint64_t top = 0;                              // doubled
        for (int k = 0; k < 2 * N; ++k) { const uint64_t nt = t[k] >> 63; t[k] = (t[k] << 1) | top; top = nt; }
        uint64_t carry = 0;                            // plus the squares
        for (int i = 0; i < N; ++i) {
            const u128 m = (u128)a.v[i] * a.v[i] + t[2 * i] + carry;
            t[2 * i] = (uint64_t)m;
            const u128 s = (u128)t[2 * i + 1] + (uint64_t)(m >> 64);
            t[2 * i + 1] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        reduce(r, t);
    }
    // a^(p - 2) (gfpInv's result for a != 0; 0 -> 0): p - 2 = 2^(64 N) - (c + 2), 4-bit windows from the top
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
        Fe<N> x = tab[15];                             // the top window of p - 2 is 1111 (c + 2 < 2^60)
        for (int bit = 64 * N - 8; bit >= 0; bit -= 4) {
            for (int k = 0; k < 4; ++k) sqr(x, x);
            const int w = (int)(e[bit >> 6] >> (bit & 63)) & 15;
            if (w) mul(x, x, tab[w]);
        }
        r = x;
    }
};

template <int N> struct Aff { Fe<N> x, y; };
template <int N> struct Jac { Fe<N> X, Y, Z; };     // (X / Z^2, Y / Z^3); Z = 0: the point at infinity

template <int N>
struct Curve {
    Field<N> F;
    uint64_t q[N];
    Aff<N> tabG[2][32];            // (2 i + 1) G and (2 i + 1) 2^(32 N) G, i = 0..31, affine
    bool ready = false;

    // ---- group law, a = -3 (what ecpDblJA3 / ecpAddJ / ecpAddAJ compute, src/math/ecp/ecp_j.c:241-299,397-590)
    void dbl(Jac<N> &R, const Jac<N> &P) const         // 3M + 5S; Z3 = 2 Y Z: infinity and points of order 2 -> infinity
    {
        Fe<N> delta, gamma, beta, alpha, t0, t1;
        F.sqr(delta, P.Z);
        F.sqr(gamma, P.Y);
        F.mul(beta, P.X, gamma);
        F.sub(t0, P.X, delta);
        F.add(t1, P.X, delta);
        F.mul(alpha, t0, t1);
        F.dbl(t0, alpha);
        F.add(alpha, alpha, t0);                        // 3 (X - Z^2)(X + Z^2)
        F.add(t0, P.Y, P.Z);
        F.sqr(t0, t0);
        F.sub(t0, t0, gamma);
        F.sub(R.Z, t0, delta);
        F.dbl(beta, beta);
        F.dbl(beta, beta);                              // 4 beta
        F.sqr(t0, alpha);
        F.dbl(t1, beta);
        F.sub(R.X, t0, t1);                             // alpha^2 - 8 beta
        F.sub(t0, beta, R.X);
        F.mul(t0, alpha, t0);
        F.sqr(gamma, gamma);
        F.dbl(gamma, gamma);
        F.dbl(gamma, gamma);
        F.dbl(gamma, gamma);                            // 8 gamma^2
        F.sub(R.Y, t0, gamma);
    }
    void from_aff(Jac<N> &R, const Aff<N> &A, bool negate) const
    {
        R.X = A.x;
        if (negate) F.neg(R.Y, A.y); else R.Y = A.y;
        memset(&R.Z, 0, sizeof R.Z);
        R.Z.v[0] = 1;
    }
    // R = P +- A (A affine, never infinity); 7M + 4S
    void madd(Jac<N> &R, const Jac<N> &P, const Aff<N> &A, bool negate) const
    {
        if (F.is_zero(P.Z)) { from_aff(R, A, negate); return; }
        Fe<N> z1z1, u2, s2, h, hh, i, j, r, v, t0, ay;
        if (negate) F.neg(ay, A.y); else ay = A.y;
        F.sqr(z1z1, P.Z);
        F.mul(u2, A.x, z1z1);
        F.mul(s2, ay, P.Z);
        F.mul(s2, s2, z1z1);
        F.sub(h, u2, P.X);
        F.sub(r, s2, P.Y);
        if (F.is_zero(h)) {
            if (F.is_zero(r)) { Jac<N> t; from_aff(t, A, negate); dbl(R, t); }       // P = A
            else memset(&R, 0, sizeof R);                                            // P = -A
            return;
        }
        F.dbl(r, r);
        F.sqr(hh, h);
        F.dbl(i, hh);
        F.dbl(i, i);
        F.mul(j, h, i);
        F.mul(v, P.X, i);
        F.add(t0, P.Z, h);
        F.sqr(t0, t0);
        F.sub(t0, t0, z1z1);
        Fe<N> z3;
        F.sub(z3, t0, hh);
        Fe<N> x3, y3;
        F.sqr(x3, r);
        F.sub(x3, x3, j);
        F.dbl(t0, v);
        F.sub(x3, x3, t0);
        F.sub(t0, v, x3);
        F.mul(y3, r, t0);
        F.mul(t0, P.Y, j);
        F.dbl(t0, t0);
        F.sub(y3, y3, t0);
        R.X = x3; R.Y = y3; R.Z = z3;
    }
    // R = P +- Q, both Jacobian; 11M + 5S
    void add(Jac<N> &R, const Jac<N> &P, const Jac<N> &Qp, bool negate) const
    {
        Jac<N> Q = Qp;
        if (negate) F.neg(Q.Y, Qp.Y);
        if (F.is_zero(P.Z)) { R = Q; return; }
        if (F.is_zero(Q.Z)) { R = P; return; }
        Fe<N> z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t0;
        F.sqr(z1z1, P.Z);
        F.sqr(z2z2, Q.Z);
        F.mul(u1, P.X, z2z2);
        F.mul(u2, Q.X, z1z1);
        F.mul(s1, P.Y, Q.Z);
        F.mul(s1, s1, z2z2);
        F.mul(s2, Q.Y, P.Z);
        F.mul(s2, s2, z1z1);
        F.sub(h, u2, u1);
        F.sub(r, s2, s1);
        if (F.is_zero(h)) {
            if (F.is_zero(r)) dbl(R, P);
            else memset(&R, 0, sizeof R);
            return;
        }
        F.dbl(r, r);
        F.dbl(i, h);
        F.sqr(i, i);
        F.mul(j, h, i);
        F.mul(v, u1, i);
        Fe<N> x3, y3, z3;
        F.add(t0, P.Z, Q.Z);
        F.sqr(t0, t0);
        F.sub(t0, t0, z1z1);
        F.sub(t0, t0, z2z2);
        F.mul(z3, t0, h);
        F.sqr(x3, r);
        F.sub(x3, x3, j);
        F.dbl(t0, v);
        F.sub(x3, x3, t0);
        F.sub(t0, v, x3);
        F.mul(y3, r, t0);
        F.mul(t0, s1, j);
        F.dbl(t0, t0);
        F.sub(y3, y3, t0);
        R.X = x3; R.Y = y3; R.Z = z3;
    }

    // c = 2^(64 N) - p, q and y_G as little-endian octet strings of 8 N octets (bign_params.c:34-140; G = (0, y_G))
    void init(uint64_t c, const uint8_t *q_le, const uint8_t *yG_le)
    {
        F.c = c;
        for (int i = 0; i < N; ++i) q[i] = hostp::ld64le(q_le + 8 * i);
        Aff<N> G;
        memset(&G, 0, sizeof G);
        for (int i = 0; i < N; ++i) G.y.v[i] = hostp::ld64le(yG_le + 8 * i);
        Jac<N> J[2][32], B, D;
        from_aff(B, G, false);
        for (int h = 0; h < 2; ++h) {
            J[h][0] = B;
            dbl(D, B);
            for (int i = 1; i < 32; ++i) add(J[h][i], J[h][i - 1], D, false);
            if (h == 0) for (int k = 0; k < 32 * N; ++k) dbl(B, B);       // 2^(32 N) G
        }
        // all 64 to affine with one inversion (no Z is zero: G has prime order q > 2^(64 N - 1))
        Fe<N> pre[64], acc, zi, zi2;
        Jac<N> *L = &J[0][0];
        pre[0] = L[0].Z;
        for (int i = 1; i < 64; ++i) F.mul(pre[i], pre[i - 1], L[i].Z);
        F.inv(acc, pre[63]);
        for (int i = 63; i >= 0; --i) {
            if (i) { F.mul(zi, acc, pre[i - 1]); F.mul(acc, acc, L[i].Z); } else zi = acc;
            F.sqr(zi2, zi);
            Aff<N> &A = tabG[i / 32][i % 32];
            F.mul(A.x, L[i].X, zi2);
            F.mul(zi2, zi2, zi);
            F.mul(A.y, L[i].Y, zi2);
        }
        ready = true;
    }
};

// width-w NAF of the nl-limb number k, least significant digit first; returns the number of digits (<= 64 nl + 1)
static inline int wnaf(int8_t *out, const uint64_t *k, int nl, int w)
{
    uint64_t t[10];
    for (int i = 0; i < nl; ++i) t[i] = k[i];
    t[nl] = 0;
    const int n = nl + 1;
    int len = 0;
    for (;;) {
        uint64_t any = 0;
        for (int i = 0; i < n; ++i) any |= t[i];
        if (!any) break;
        int d = 0;
        if (t[0] & 1) {
            d = (int)(t[0] & ((1u << w) - 1));
            if (d >= (1 << (w - 1))) d -= 1 << w;
            if (d > 0) {
                u128 s = (u128)t[0] - (uint64_t)d;
                t[0] = (uint64_t)s;
                for (int i = 1; i < n && ((uint64_t)(s >> 64) & 1); ++i) { s = (u128)t[i] - 1; t[i] = (uint64_t)s; }
            } else {
                u128 s = (u128)t[0] + (uint64_t)(-d);
                t[0] = (uint64_t)s;
                for (int i = 1; i < n && (uint64_t)(s >> 64); ++i) { s = (u128)t[i] + 1; t[i] = (uint64_t)s; }
            }
        }
        out[len++] = (int8_t)d;
        for (int i = 0; i + 1 < n; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 63);
        t[n - 1] >>= 1;
    }
    return len;
}

static inline int cmp_limbs(const uint64_t *a, const uint64_t *b, int n)
{
    for (int i = n - 1; i >= 0; --i)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
static inline uint64_t sub_limbs(uint64_t *r, const uint64_t *a, const uint64_t *b, int n)
{
    uint64_t borrow = 0;
    for (int i = 0; i < n; ++i) {
        const u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

// belt-hash of a message given in pieces (beltHashStart / StepH / StepG, belt_hash.c:28-135) on hostp::hash_stream
struct BeltHashPieces {
    const hostp::BeltTables &T;
    uint32_t hs[12];
    uint8_t block[32];
    size_t filled = 0;
    uint64_t bits_lo = 0, bits_hi = 0;
    BeltHashPieces(const hostp::BeltTables &t, const uint8_t H[256]) : T(t)
    {
        for (int i = 0; i < 8; ++i) hs[i] = hostp::ld32le(H + 4 * i);
        for (int i = 8; i < 12; ++i) hs[i] = 0;
    }
    void absorb(const uint8_t *p, size_t count)
    {
        const uint64_t add = (uint64_t)count << 3;
        bits_lo += add;
        bits_hi += ((uint64_t)count >> 61) + (bits_lo < add);
        if (filled) {
            size_t take = 32 - filled;
            if (take > count) take = count;
            memcpy(block + filled, p, take);
            filled += take; p += take; count -= take;
            if (filled < 32) return;
            hostp::hash_stream(T, hs, block, 1, 0, 0, 0);
            filled = 0;
        }
        const size_t full = count / 32;
        if (full) hostp::hash_stream(T, hs, p, full, 0, 0, 0);
        if (count % 32) { memcpy(block, p + 32 * full, count % 32); filled = count % 32; }
    }
    void digest(uint8_t out[32])
    {
        size_t n = 0;
        if (filled) { memset(block + filled, 0, 32 - filled); n = 1; }
        hostp::hash_stream(T, hs, block, n, 1, bits_lo, bits_hi);
        for (int i = 0; i < 8; ++i) hostp::st32le(out + 4 * i, hs[i]);
    }
};

// bignVerifyEc (bign_sign.c:268-344) for the curve E of level l = 32 N: hash[8 N], sig[12 N] = s0 || s1, pubkey[16 N].
// The caller has validated the parameters and the OID (bign_sign.c:288-292,355-358).  rx_out (8 N octets, may be null)
// receives <x_R> when R != O (tests).
template <int N>
static inline uint32_t verify(const Curve<N> &E, const hostp::BeltTables &T, const uint8_t H[256], const uint8_t *oid_der,
                              size_t oid_len, const uint8_t *hash, const uint8_t *sig, const uint8_t *pubkey,
                              uint8_t *rx_out = nullptr)
{
    const Field<N> &F = E.F;
    constexpr int no = 8 * N, HN = N / 2;
    // Q: both coordinates below p (qrFrom, :306-311); no on-curve test at this point of the reference either
    Aff<N> Q;
    for (int i = 0; i < N; ++i) { Q.x.v[i] = hostp::ld64le(pubkey + 8 * i); Q.y.v[i] = hostp::ld64le(pubkey + no + 8 * i); }
    if (F.ge_p(Q.x) || F.ge_p(Q.y)) return kBadPubkey;
    // s1 < q (:313-318)
    uint64_t s1[N], h[N], u[N];
    for (int i = 0; i < N; ++i) { s1[i] = hostp::ld64le(sig + no / 2 + 8 * i); h[i] = hostp::ld64le(hash + 8 * i); }
    if (cmp_limbs(s1, E.q, N) >= 0) return kBadSig;
    // u = (s1 + H) mod q, H first reduced by one subtraction (:320-327)
    if (cmp_limbs(h, E.q, N) >= 0) sub_limbs(h, h, E.q, N);
    {
        uint64_t carry = 0;
        for (int i = 0; i < N; ++i) { const u128 s = (u128)s1[i] + h[i] + carry; u[i] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
        if (carry || cmp_limbs(u, E.q, N) >= 0) sub_limbs(u, u, E.q, N);
    }
    // v = s0 + 2^l (:329-330)
    uint64_t v[HN + 1];
    for (int i = 0; i < HN; ++i) v[i] = hostp::ld64le(sig + 8 * i);
    v[HN] = 1;
    // digits
    int8_t d0[32 * N + 2], d1[32 * N + 2], dv[32 * N + 3];
    const int n0 = wnaf(d0, u, HN, 7), n1 = wnaf(d1, u + HN, HN, 7), nv = wnaf(dv, v, HN + 1, 5);
    // odd multiples 1..15 of Q, Jacobian
    Jac<N> TQ[8], D;
    E.from_aff(TQ[0], Q, false);
    E.dbl(D, TQ[0]);
    for (int i = 1; i < 8; ++i) E.add(TQ[i], TQ[i - 1], D, false);
    // R = u0 G + u1 (2^l G) + v Q (:332)
    Jac<N> R;
    memset(&R, 0, sizeof R);
    int top = nv;
    if (n0 > top) top = n0;
    if (n1 > top) top = n1;
    for (int k = top - 1; k >= 0; --k) {
        E.dbl(R, R);
        if (k < nv && dv[k]) { const int d = dv[k]; E.add(R, R, TQ[(d < 0 ? -d : d) >> 1], d < 0); }
        if (k < n0 && d0[k]) { const int d = d0[k]; E.madd(R, R, E.tabG[0][(d < 0 ? -d : d) >> 1], d < 0); }
        if (k < n1 && d1[k]) { const int d = d1[k]; E.madd(R, R, E.tabG[1][(d < 0 ? -d : d) >> 1], d < 0); }
    }
    if (F.is_zero(R.Z)) return kBadSig;                 // ecAddMulA returned FALSE (:332-336)
    Fe<N> zi, x;
    F.inv(zi, R.Z);
    F.sqr(zi, zi);
    F.mul(x, R.X, zi);
    uint8_t rx[no], t[32];
    for (int i = 0; i < N; ++i) hostp::st64le(rx + 8 * i, x.v[i]);
    if (rx_out) memcpy(rx_out, rx, no);
    // s0 == belt-hash(oid || <x_R> || H) mod 2^l ? (:337-343)
    BeltHashPieces bh(T, H);
    bh.absorb(oid_der, oid_len);
    bh.absorb(rx, no);
    bh.absorb(hash, no);
    bh.digest(t);
    return memcmp(t, sig, no / 2) == 0 ? kOk : kBadSig;
}

// bignPubkeyVal (bign_misc.c:319-365) for the curve of level l = 32 N: both coordinates below p (qrFrom), then
// ecpIsOnA (src/math/ecp/ecp_a.c:36-60): (x^2 + a) x + b == y^2 with a = p - 3; b_le = b as 8 N little-endian octets
template <int N>
static inline uint32_t pubkey_val(const Curve<N> &E, const uint8_t *b_le, const uint8_t *pubkey)
{
    const Field<N> &F = E.F;
    Fe<N> x, y, b, t, three;
    for (int i = 0; i < N; ++i) {
        x.v[i] = hostp::ld64le(pubkey + 8 * i);
        y.v[i] = hostp::ld64le(pubkey + 8 * N + 8 * i);
        b.v[i] = hostp::ld64le(b_le + 8 * i);
    }
    if (F.ge_p(x) || F.ge_p(y)) return kBadPubkey;
    memset(&three, 0, sizeof three);
    three.v[0] = 3;
    F.sqr(t, x);
    F.sub(t, t, three);
    F.mul(t, t, x);
    F.add(t, t, b);
    F.sqr(y, y);
    return Field<N>::eq(t, y) ? kOk : kBadPubkey;
}

}  // namespace hostb
}  // namespace bee2hip
