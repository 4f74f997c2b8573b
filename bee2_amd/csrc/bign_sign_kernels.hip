// bign_sign_kernels.hip -- batched public-key calculation, key generation and signing on the three standard
// bign curves (gfx950).  SURVEY.md 8f-4, second half.  Replaces n calls of
//   bignPubkeyCalc  src/crypto/bign/bign_misc.c:373-417   (0 < d < q, Q = d G)
//   bignKeypairGen  src/crypto/bign/bign_misc.c:182-229   (d from the caller's rng on the host, Q = d G here)
//   bignSign        src/crypto/bign/bign_sign.c:32-112    (k from the caller's rng on the host)
//   bignSign2       src/crypto/bign/bign_sign.c:140-245   (k by STB 34.101.45 algorithm 6.3.3, on the device)
// and the bign128 / bign192 / bign256 facades of each.  One lane per key / signature.
//
// Everything below handles SECRETS (d, k) and is written to be constant-time at the instruction level:
//   * no branch condition and no global / scalar memory address depends on a secret.  Loops run over public
//     counts; selections are masks (v_bitop3 / v_cndmask), never control flow.  The one exception is the
//     rejection loop "k <- belt-wbl(k) until 0 < k < q" of algorithm 6.3.3 (bign_sign.c:206-216), which the
//     reference has too and which repeats with probability (2^2l - q) / 2^2l < 2^-126 per signature;
//   * the reference's bignMulBase (src/crypto/bign/bign_lcl.c, tables under src/crypto/bign/pre) reads a
//     precomputed table at secret indices with a regular recoding.  Here k G is a fixed-base comb over 4-bit
//     windows (8N windows, no doublings) whose table row -- 15 affine multiples of 16^w G -- is read IN FULL
//     for every window with wavefront-uniform scalar loads (s_load_dwordx16: the addresses depend on the
//     window number only) and the wanted entry is picked with masks;
//   * the additions use the COMPLETE projective formulas of Renes, Costello and Batina (algorithm 5: a = -3,
//     mixed, 11 M + 2 m_b), so there is no exceptional case to branch on: k = q - 2 d_0, acc = O, digit = 0
//     all go through the same instructions (a zero digit adds a dummy point and keeps the old accumulator);
//   * the inversion for the affine result is by division steps with a FIXED number of iterations (fe_inv_safegcd<N, true>;
//     round 2 first used a^(p-2), a fixed chain of squarings and multiplications four times as long),
//     where the verify path's variant of the same function leaves its loop as soon as g = 0 in the whole wavefront;
//   * belt (theta = belt-hash(oid || d || t), k = belt-wbl_theta(H)) looks its S-box up in LDS at secret
//     indices, as every table-driven belt does.  The table used here is BeltTabTwo, whose copies are
//     bank-private (lane l only ever touches bank l mod 32, belt_dev.hpp): an access takes the same LDS
//     cycles whatever the index (SQ_LDS_BANK_CONFLICT = 0), so the lookups leak nothing through timing;
//   * s1 = (k - (s0 + 2^l) d - H) mod q: schoolbook product, Crandall-style folds by 2^2l - q and masked
//     conditional subtractions.
// profiles/r02_sign_ct_audit.txt lists every conditional branch of these kernels with what it tests.
//
// Kernels: pubkey calc = mulbase<CHECK_D>;  sign2 = nonce -> mulbase -> tail;  sign with given k = kcheck -> mulbase -> tail.
#include <mutex>
#include "belt_dev.hpp"
#include "bign_dev.hpp"
#include "bign_fe29.hpp"
#include "common.hpp"

namespace bee2hip {

constexpr uint32_t ERR_BAD_PRIVKEY_V = 504;          // err.h:184
constexpr uint32_t ERR_BAD_RNG_V = 304;              // err.h:138

__constant__ uint32_t c_cq8[5] = BIGN128_CQ_LIMBS;   // 2^2l - q, N/2 + 1 limbs
__constant__ uint32_t c_cq12[7] = BIGN192_CQ_LIMBS;
__constant__ uint32_t c_cq16[9] = BIGN256_CQ_LIMBS;
template <int N> __device__ __forceinline__ const uint32_t *curve_cq() { return N == 8 ? c_cq8 : N == 12 ? c_cq12 : c_cq16; }

// ------------------------------------------------------------ mask helpers ---
// all-ones iff a == b, for values below 2^31
__device__ __forceinline__ uint32_t ct_eq_small(uint32_t a, uint32_t b) { return (uint32_t)((int32_t)((a ^ b) - 1u) >> 31); }
// all-ones iff x == 0 (any 32-bit x)
__device__ __forceinline__ uint32_t ct_is_zero32(uint32_t x) { return (uint32_t)((int32_t)(~x & (x - 1u)) >> 31); }
// r = m ? a : b, m all-ones or zero
__device__ __forceinline__ uint32_t ct_sel(uint32_t m, uint32_t a, uint32_t b) { return bitop3<0xCA>(m, a, b); }   // (m & a) | (~m & b)
template <int N>
__device__ __forceinline__ uint32_t ct_is_zero(const uint32_t (&a)[N])
{
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z |= a[i];
    return ct_is_zero32(z);
}
// all-ones iff a < b (multi-limb, raw values)
template <int N>
__device__ __forceinline__ uint32_t ct_lt(const uint32_t (&a)[N], const uint32_t (&b)[N])
{
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        borrow = (uint32_t)(d >> 32) & 1u;
    }
    return 0u - borrow;
}
// all-ones iff 0 < x < q
template <int N>
__device__ __forceinline__ uint32_t ct_in_range_q(const uint32_t (&x)[N])
{
    uint32_t q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = curve_q<N>()[i];
    return ~ct_is_zero(x) & ct_lt(x, q);
}
// the group order as a launch argument: the nonce / range-check kernels serve any parameter set (round 3: signing on
// non-standard sets, bign_generic_kernels.hip); std = 1 means "the standard set of this level" (q from constant memory)
template <int N> struct QArg { uint32_t q[N]; uint32_t std; };
template <int N>
__device__ __forceinline__ uint32_t ct_in_range_q(const uint32_t (&x)[N], const QArg<N> &qa)
{
    uint32_t q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = qa.std ? curve_q<N>()[i] : qa.q[i];       // qa.std is wavefront-uniform and public
    return ~ct_is_zero(x) & ct_lt(x, q);
}

// ------------------------------------------------- complete mixed addition ---
// P <- P + Q on y^2 = x^3 - 3x + b, P = (X : Y : Z) homogeneous projective (x = X / Z, O = (0 : 1 : 0)),
// Q = (x2, y2) affine and on the curve.  Renes-Costello-Batina, "Complete addition formulas for prime order
// elliptic curves", algorithm 5: valid for EVERY P including O, P = Q and P = -Q (the group order is odd).
// Checked against affine arithmetic incl. those cases before it was written (tools/model_sign.py).
template <int N> struct projT { feT<N> X, Y, Z; };
template <int N>
__device__ __forceinline__ void proj_madd_complete(projT<N> &P, const affT<N> &Q, const feT<N> &b)
{
    feT<N> t0, t1, t2, t3, t4, X3, Y3, Z3;
    fe_mul(t0, P.X, Q.x);
    fe_mul(t1, P.Y, Q.y);
    fe_add(t3, Q.x, Q.y);
    fe_add(t4, P.X, P.Y);
    fe_mul(t3, t3, t4);
    fe_add(t4, t0, t1);
    fe_sub(t3, t3, t4);
    fe_mul(t4, Q.y, P.Z);
    fe_add(t4, t4, P.Y);
    fe_mul(Y3, Q.x, P.Z);
    fe_add(Y3, Y3, P.X);
    fe_mul(Z3, b, P.Z);
    fe_sub(X3, Y3, Z3);
    fe_add(Z3, X3, X3);
    fe_add(X3, X3, Z3);
    fe_sub(Z3, t1, X3);
    fe_add(X3, t1, X3);
    fe_mul(Y3, b, Y3);
    fe_add(t1, P.Z, P.Z);
    fe_add(t2, t1, P.Z);
    fe_sub(Y3, Y3, t2);
    fe_sub(Y3, Y3, t0);
    fe_add(t1, Y3, Y3);
    fe_add(Y3, t1, Y3);
    fe_add(t1, t0, t0);
    fe_add(t0, t1, t0);
    fe_sub(t0, t0, t2);
    fe_mul(t1, t4, Y3);
    fe_mul(t2, t0, Y3);
    fe_mul(Y3, X3, Z3);
    fe_add(Y3, Y3, t2);
    fe_mul(X3, t3, X3);
    fe_sub(P.X, X3, t1);
    fe_mul(Z3, t4, Z3);
    fe_mul(t1, t3, t0);
    fe_add(P.Z, Z3, t1);
    P.Y = Y3;
}

template <int N>
__device__ __forceinline__ uint32_t proj_to_affine_ct(feT<N> &x, feT<N> &y, const projT<N> &acc);

// -------------------------------------------------------------- k G, affine ---
// R = k G for a secret k (any N-limb value; the callers have k in {1 .. q - 1}).  gtab8 is the verify path's 8-bit
// seed table: entry (win, b) = b 2^(8 win) G affine, 2N words, at index win * 256 + b; window w of 4 bits reads
// its 15 entries j 16^w G = entry (w / 2, j << 4 (w & 1)).
template <int N>
__device__ __forceinline__ uint32_t mul_base_ct(feT<N> &x, feT<N> &y, const uint32_t (&k)[N], const uint32_t *__restrict__ gtab8)
{
    uint32_t kk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) kk[i] = k[i];
    feT<N> b;
#pragma unroll
    for (int i = 0; i < N; ++i) b.v[i] = curve_b<N>()[i];
    projT<N> acc;
    fe_set_zero(acc.X); fe_set_one(acc.Y); fe_set_zero(acc.Z);          // O
#pragma unroll 1
    for (int w = 0; w < 8 * N; ++w) {
        const uint32_t dg = kk[0] & 15u;
        // k >>= 4
#pragma unroll
        for (int i = 0; i < N - 1; ++i) kk[i] = __builtin_amdgcn_alignbit(kk[i + 1], kk[i], 4);
        kk[N - 1] >>= 4;
        // scan the whole row of the table; addresses depend on w only (scalar loads)
        const uint32_t *row = gtab8 + ((size_t)(w >> 1) * GT8_ENTRIES) * (2 * N);
        const int sh = 4 * (w & 1);
        affT<N> E;
        fe_set_zero(E.x); fe_set_zero(E.y);
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            const uint32_t m = ct_eq_small(dg, (uint32_t)j);
            const uint32_t *e = row + (size_t)(j << sh) * (2 * N);
#pragma unroll
            for (int l = 0; l < N; ++l) {
                E.x.v[l] = bitop3<0xF8>(E.x.v[l], e[l], m);             // a | (b & c)
                E.y.v[l] = bitop3<0xF8>(E.y.v[l], e[N + l], m);
            }
        }
        // digit 0: E = (0, 0), not a point -- add it all the same and keep the old accumulator
        projT<N> sum = acc;
        proj_madd_complete(sum, E, b);
        const uint32_t keep = ct_eq_small(dg, 0u);
#pragma unroll
        for (int l = 0; l < N; ++l) {
            acc.X.v[l] = ct_sel(keep, acc.X.v[l], sum.X.v[l]);
            acc.Y.v[l] = ct_sel(keep, acc.Y.v[l], sum.Y.v[l]);
            acc.Z.v[l] = ct_sel(keep, acc.Z.v[l], sum.Z.v[l]);
        }
    }
    return proj_to_affine_ct(x, y, acc);
}

// affine x = X / Z, y = Y / Z of a secret point; all-ones iff the point is O (Z = 0: k = 0 mod q) -- then x = y = 0
template <int N>
__device__ __forceinline__ uint32_t proj_to_affine_ct(feT<N> &x, feT<N> &y, const projT<N> &acc)
{
    feT<N> zc;
    fe_canon(zc, acc.Z);
    // division steps with a fixed iteration count (bign_dev.hpp, fe_inv_safegcd<N, true>): a quarter of the
    // instructions of the a^(p-2) chain, which was a fifth of this kernel.  The product Z * Z^-1 is checked with masks;
    // a failure (never observed; the bound on the number of steps is proven) is reported like k G = O, i.e. loudly.
    const feT<N> zi = fe_inv_safegcd<N, true>(zc);
    feT<N> chk;
    fe_mul(chk, zc, zi);
    fe_canon(chk, chk);
    chk.v[0] ^= 1u;                                // 0 iff Z * Z^-1 == 1
    fe_mul(x, acc.X, zi);
    fe_mul(y, acc.Y, zi);
    fe_canon(x, x);
    fe_canon(y, y);
    return ct_is_zero(zc.v) | ~ct_is_zero(chk.v);  // all-ones iff k G = O (or the inversion check failed)
}

// The same with SIGNED 6-bit windows (round 3): W6 = ceil((32N + 1) / 6) complete additions instead of 8N (43 / 65 / 86
// against 64 / 96 / 128), each after a masked scan of the window's 32 table entries |d| 2^(6w) G (gtab6, bign_kernels.hip
// bign_gtable6_kernel) and a masked negation of y.  The digits come out of the scalar low to high with the usual carry:
// t = bits + carry in [0, 64], d = t - 64 [t >= 32] in [-32, 31]; the last window holds what is left of k plus the carry
// (<= 16 / 1 / 4), never negative, so no carry leaves it.  Nothing but masks touches d.
// JAC: the accumulator in Jacobian coordinates and the mixed addition 8M + 3S (madd-2007-bl shape, as jac_madd of the
// verification side, written here without its flags) instead of the complete 11M + 2 m_b one.  The formula is not complete, the
// SCHEDULE makes it safe: before window w the accumulator is A G with |A| <= 32 (64^w - 1) / 63 < 64^w <= |d_w 64^w|, all below
// q / 2 up to the last window, so A = +-d_w 64^w (mod q) cannot happen and A = 0 only while every digit so far was 0 -- that
// state is carried as a mask (the first non-zero digit SETS the accumulator).  In the last window d 64^w may pass q; the one
// collision is k = 0 (mod q), where H = 0, r != 0 and the formula returns Z3 = Z1 H = 0: the point at infinity, which is the
// right answer and is reported.  (k = q is reachable in key generation only, which takes any d below 2^(2l).)
template <int N>
__device__ __forceinline__ void jac_madd_ct(jacT<N> &T, const affT<N> &E)
{
    feT<N> Z1Z1, U2, S2, H, HH, HHH, r, V, t;
    fe_sqr(Z1Z1, T.Z);
    fe_mul(U2, E.x, Z1Z1);
    fe_mul(t, T.Z, Z1Z1);
    fe_mul(S2, E.y, t);
    fe_sub(H, U2, T.X);
    fe_sub(r, S2, T.Y);
    fe_sqr(HH, H);
    fe_mul(HHH, H, HH);
    fe_mul(V, T.X, HH);
    fe_mul(T.Z, T.Z, H);                                // Z3 = Z1 H
    fe_sqr(t, r);
    fe_sub(t, t, HHH);
    fe_dbl(U2, V);
    fe_sub(T.X, t, U2);                                 // X3 = r^2 - H^3 - 2 V
    fe_sub(t, V, T.X);
    fe_mul(t, r, t);
    fe_mul(S2, T.Y, HHH);
    fe_sub(T.Y, t, S2);                                 // Y3 = r (V - X3) - Y1 H^3
}

template <int N, bool JAC>
__device__ __forceinline__ uint32_t mul_base_ct6(feT<N> &x, feT<N> &y, const uint32_t (&k)[N], const uint32_t *__restrict__ gtab6)
{
    uint32_t kk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) kk[i] = k[i];
    feT<N> b;
#pragma unroll
    for (int i = 0; i < N; ++i) b.v[i] = curve_b<N>()[i];
    projT<N> acc;
    fe_set_zero(acc.X); fe_set_one(acc.Y); fe_set_zero(acc.Z);          // O
    jacT<N> J;
    fe_set_zero(J.X); fe_set_one(J.Y); fe_set_zero(J.Z);
    uint32_t at_inf = ~0u;                                              // JAC: all-ones while no non-zero digit has been met
    feT<N> one;
    fe_set_one(one);
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < Win6<N>::W; ++w) {
        const uint32_t t = (kk[0] & 63u) + carry;                       // 0 .. 64
#pragma unroll
        for (int i = 0; i < N - 1; ++i) kk[i] = __builtin_amdgcn_alignbit(kk[i + 1], kk[i], 6);
        kk[N - 1] >>= 6;
        carry = (t + 32u) >> 6;                                         // 1 iff t >= 32
        const uint32_t d = t - (carry << 6);                            // two's complement of the digit
        const uint32_t neg = (uint32_t)((int32_t)d >> 31);              // all-ones iff d < 0
        const uint32_t mag = (d ^ neg) - neg;                           // |d| in 0 .. 32
        const uint32_t *row = gtab6 + (size_t)w * 32 * (2 * N);         // depends on w only: scalar loads
        affT<N> E;
        fe_set_zero(E.x); fe_set_zero(E.y);
#pragma unroll 8
        for (int j = 1; j <= 32; ++j) {
            const uint32_t m = ct_eq_small(mag, (uint32_t)j);
            const uint32_t *e = row + (size_t)(j - 1) * (2 * N);
#pragma unroll
            for (int l = 0; l < N; ++l) {
                E.x.v[l] = bitop3<0xF8>(E.x.v[l], e[l], m);             // a | (b & c)
                E.y.v[l] = bitop3<0xF8>(E.y.v[l], e[N + l], m);
            }
        }
        feT<N> ny;
        fe_neg(ny, E.y);                                                // p - y (p for the dummy entry of digit 0)
#pragma unroll
        for (int l = 0; l < N; ++l) E.y.v[l] = ct_sel(neg, ny.v[l], E.y.v[l]);
        const uint32_t keep = ct_eq_small(mag, 0u);
        if constexpr (JAC) {
            jacT<N> sum = J;
            jac_madd_ct(sum, E);                                        // digit 0 or accumulator still O: computed, not used
            const uint32_t set = at_inf & ~keep;                        // first non-zero digit: J <- (x, y, 1)
#pragma unroll
            for (int l = 0; l < N; ++l) {
                J.X.v[l] = ct_sel(keep, J.X.v[l], ct_sel(set, E.x.v[l], sum.X.v[l]));
                J.Y.v[l] = ct_sel(keep, J.Y.v[l], ct_sel(set, E.y.v[l], sum.Y.v[l]));
                J.Z.v[l] = ct_sel(keep, J.Z.v[l], ct_sel(set, one.v[l], sum.Z.v[l]));
            }
            at_inf &= keep;
        } else {
            projT<N> sum = acc;
            proj_madd_complete(sum, E, b);                              // digit 0: (0, 0) is not a point; the old accumulator is kept
#pragma unroll
            for (int l = 0; l < N; ++l) {
                acc.X.v[l] = ct_sel(keep, acc.X.v[l], sum.X.v[l]);
                acc.Y.v[l] = ct_sel(keep, acc.Y.v[l], sum.Y.v[l]);
                acc.Z.v[l] = ct_sel(keep, acc.Z.v[l], sum.Z.v[l]);
            }
        }
    }
    if constexpr (JAC) {
        // x = X / Z^2, y = Y / Z^3; Z = 0 (or nothing ever added: k = 0) is the point at infinity
#pragma unroll
        for (int l = 0; l < N; ++l) J.Z.v[l] &= ~at_inf;
        feT<N> zc;
        fe_canon(zc, J.Z);
        const feT<N> zi = fe_inv_safegcd<N, true>(zc);
        feT<N> chk, zi2;
        fe_mul(chk, zc, zi);
        fe_canon(chk, chk);
        chk.v[0] ^= 1u;                                                 // 0 iff Z * Z^-1 == 1
        fe_sqr(zi2, zi);
        fe_mul(x, J.X, zi2);
        fe_mul(zi2, zi2, zi);
        fe_mul(y, J.Y, zi2);
        fe_canon(x, x);
        fe_canon(y, y);
        const uint32_t inf = ct_is_zero(zc.v) | ~ct_is_zero(chk.v);
#pragma unroll
        for (int l = 0; l < N; ++l) { x.v[l] &= ~inf; y.v[l] &= ~inf; }
        return inf;
    } else {
        return proj_to_affine_ct(x, y, acc);
    }
}

// ---- the same multiplication with the window's entry LOOKED UP in LDS (round 4) ---------------------------------------------
// The masked scan above costs one full-rate VALU op per table word per entry: 512 + 96 of the ~3 800 instructions of a
// window, and it is what kept the windows at 6 bits.  Here the 1024 lanes of a workgroup walk the windows together and the
// window's row sits in LDS in 32 copies, copy r holding its 64-bit words at byte address (word * 32 + r) * 8: lane l reads
// words (entry * N + i) of copy l mod 32, i.e. ALWAYS the two banks 2 (l mod 32), 2 (l mod 32) + 1 of the 64
// (ds_read_b64: lane groups {0-31}, {32-63}, bank = (a / 4) mod 64, MI355X_MICROARCH.md LDS table).  Whatever the secret
// entry numbers are, the 32 lanes of a group hit 32 disjoint bank pairs: no conflict, the same LDS cycles for every scalar --
// the argument of the bank-private belt tables (belt_dev.hpp), with SQ_LDS_BANK_CONFLICT = 0 to show for it
// (profiles/r04_sign_lds.txt).  A row of 2^(WB-1) entries x 8N octets x 32 copies must fit the CU's 160 KiB: signed 7-bit
// windows on the 256-bit curve (64 entries: 128 KiB, one workgroup of 1024 lanes = 4 wavefronts per SIMD, 37 additions instead
// of 43).  Per window: barrier, 16 broadcast loads + ds_write_b64 per lane to refill the row (addresses depend on the window
// and the lane only), barrier, N ds_read_b64 -- ~60 instructions where the scan took ~730.  Digits, negation, the masked
// Jacobian addition and the schedule argument are those of mul_base_ct6<N, true>: digits t - 128 [t >= 64] in [-64, 63],
// |A| <= 64 (128^w - 1) / 127 < 128^w before window w, and the top window starts at bit 252 exactly as with 6-bit windows.
template <int N, int WB>
__device__ __forceinline__ uint32_t mul_base_ct_lds(feT<N> &x, feT<N> &y, const uint32_t (&k)[N], const uint64_t *__restrict__ tabw,
                                                    uint64_t *s_row, const int x_only /* public: signing needs x_R alone */)
{
    constexpr int W = WinW<N, WB>::W, ENT = WinW<N, WB>::ENT, QW = ENT * N;       // 64-bit words of a row
    static_assert(QW * 32 % 1024 == 0, "the refill is written for 1024 lanes");
    const unsigned tid = threadIdx.x, rep = tid & 31u;
    uint32_t kk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) kk[i] = k[i];
    jacT<N> J;
    fe_set_zero(J.X); fe_set_one(J.Y); fe_set_zero(J.Z);
    uint32_t at_inf = ~0u;
    feT<N> one;
    fe_set_one(one);
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        __syncthreads();                                                // every lane is through with the previous row
        {
            const uint64_t *row = tabw + (size_t)w * QW;
#pragma unroll
            for (int i = 0; i < QW * 32 / 1024; ++i) {
                const unsigned q = (tid >> 5) + 32u * i;                // the 32 lanes of a copy-group fetch the same word
                s_row[q * 32u + rep] = row[q];
            }
        }
        __syncthreads();
        const uint32_t t = (kk[0] & (uint32_t)(2 * ENT - 1)) + carry;   // 0 .. 2 ENT
#pragma unroll
        for (int i = 0; i < N - 1; ++i) kk[i] = __builtin_amdgcn_alignbit(kk[i + 1], kk[i], WB);
        kk[N - 1] >>= WB;
        carry = (t + (uint32_t)ENT) >> WB;                              // 1 iff t >= ENT
        const uint32_t d = t - (carry << WB);                           // two's complement of the digit
        const uint32_t neg = (uint32_t)((int32_t)d >> 31);
        const uint32_t mag = (d ^ neg) - neg;                           // 0 .. ENT
        const uint32_t e = (mag - 1u) & (uint32_t)(ENT - 1);            // digit 0 reads entry ENT: looked up, added, not used
        affT<N> E;
#pragma unroll
        for (int l = 0; l < N / 2; ++l) {
            const uint64_t vx = s_row[(e * N + l) * 32u + rep], vy = s_row[(e * N + N / 2 + l) * 32u + rep];
            E.x.v[2 * l] = (uint32_t)vx; E.x.v[2 * l + 1] = (uint32_t)(vx >> 32);
            E.y.v[2 * l] = (uint32_t)vy; E.y.v[2 * l + 1] = (uint32_t)(vy >> 32);
        }
        feT<N> ny;
        fe_neg(ny, E.y);
#pragma unroll
        for (int l = 0; l < N; ++l) E.y.v[l] = ct_sel(neg, ny.v[l], E.y.v[l]);
        const uint32_t keep = ct_eq_small(mag, 0u);
        jacT<N> sum = J;
        jac_madd_ct(sum, E);                                            // digit 0 or accumulator still O: computed, not used
        const uint32_t set = at_inf & ~keep;                            // first non-zero digit: J <- (x, y, 1)
#pragma unroll
        for (int l = 0; l < N; ++l) {
            J.X.v[l] = ct_sel(keep, J.X.v[l], ct_sel(set, E.x.v[l], sum.X.v[l]));
            J.Y.v[l] = ct_sel(keep, J.Y.v[l], ct_sel(set, E.y.v[l], sum.Y.v[l]));
            J.Z.v[l] = ct_sel(keep, J.Z.v[l], ct_sel(set, one.v[l], sum.Z.v[l]));
        }
        at_inf &= keep;
    }
    // x = X / Z^2, y = Y / Z^3; Z = 0 (or nothing ever added: k = 0) is the point at infinity
#pragma unroll
    for (int l = 0; l < N; ++l) J.Z.v[l] &= ~at_inf;
    feT<N> zc;
    fe_canon(zc, J.Z);
    const feT<N> zi = fe_inv_safegcd<N, true>(zc);
    feT<N> chk, zi2;
    fe_mul(chk, zc, zi);
    fe_canon(chk, chk);
    chk.v[0] ^= 1u;                                                     // 0 iff Z * Z^-1 == 1
    fe_sqr(zi2, zi);
    fe_mul(x, J.X, zi2);
    fe_canon(x, x);
    fe_set_zero(y);
    if (!x_only) {
        fe_mul(zi2, zi2, zi);
        fe_mul(y, J.Y, zi2);
        fe_canon(y, y);
    }
    const uint32_t inf = ct_is_zero(zc.v) | ~ct_is_zero(chk.v);
#pragma unroll
    for (int l = 0; l < N; ++l) { x.v[l] &= ~inf; y.v[l] &= ~inf; }
    return inf;
}

template <int N>
__device__ __forceinline__ void load_words_bytes(uint32_t (&r)[N], const uint8_t *p)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = w[i];
}

// ---------------------------------------------- k G by a whole wavefront ---
// P <- P + Q, both homogeneous projective, a = -3: Renes-Costello-Batina algorithm 4 (12M + 2 m_b + 29a), complete like the
// mixed form above.  tools/model_rcb_a3_full.py checks this operation sequence against affine arithmetic (every pair of
// points of a small odd-order curve, the standard curves, and the schedule below).
template <int N>
__device__ __forceinline__ void proj_add_complete(projT<N> &P, const projT<N> &Q, const feT<N> &b)
{
    feT<N> t0, t1, t2, t3, t4, X3, Y3, Z3;
    fe_mul(t0, P.X, Q.X);
    fe_mul(t1, P.Y, Q.Y);
    fe_mul(t2, P.Z, Q.Z);
    fe_add(t3, P.X, P.Y);
    fe_add(t4, Q.X, Q.Y);
    fe_mul(t3, t3, t4);
    fe_add(t4, t0, t1);
    fe_sub(t3, t3, t4);                 // X1 Y2 + X2 Y1
    fe_add(t4, P.Y, P.Z);
    fe_add(X3, Q.Y, Q.Z);
    fe_mul(t4, t4, X3);
    fe_add(X3, t1, t2);
    fe_sub(t4, t4, X3);                 // Y1 Z2 + Y2 Z1
    fe_add(X3, P.X, P.Z);
    fe_add(Y3, Q.X, Q.Z);
    fe_mul(X3, X3, Y3);
    fe_add(Y3, t0, t2);
    fe_sub(Y3, X3, Y3);                 // X1 Z2 + X2 Z1
    fe_mul(Z3, b, t2);
    fe_sub(X3, Y3, Z3);
    fe_add(Z3, X3, X3);
    fe_add(X3, X3, Z3);
    fe_sub(Z3, t1, X3);
    fe_add(X3, t1, X3);
    fe_mul(Y3, b, Y3);
    fe_add(t1, t2, t2);
    fe_add(t2, t1, t2);
    fe_sub(Y3, Y3, t2);
    fe_sub(Y3, Y3, t0);
    fe_add(t1, Y3, Y3);
    fe_add(Y3, t1, Y3);
    fe_add(t1, t0, t0);
    fe_add(t0, t1, t0);
    fe_sub(t0, t0, t2);
    fe_mul(t1, t4, Y3);
    fe_mul(t2, t0, Y3);
    fe_mul(Y3, X3, Z3);
    fe_add(Y3, Y3, t2);
    fe_mul(X3, t3, X3);
    fe_sub(P.X, X3, t1);
    fe_mul(Z3, t4, Z3);
    fe_mul(t1, t3, t0);
    fe_add(P.Z, Z3, t1);
    P.Y = Y3;
}

// R = k G for ONE secret k by LANES adjacent lanes of a wavefront (small batches: with one lane per scalar a lone wavefront
// walks 8N dependent additions, 0.6 ms whatever the batch).  Lane j of the group takes the 4-bit windows j, j + LANES, ...:
// the same masked scan of each window's 15 table entries as mul_base_ct (the addresses depend on the lane, not on k), then a
// butterfly of log2(LANES) complete additions leaves the sum in every lane of the group.  No branch, address or shuffle
// pattern depends on k.  LANES = 64 is the latency form (one window per lane on the 256-bit curve); 16 and 4 trade depth for
// work and fill the device at 2^12 and 2^14 scalars (profiles/r03_sign_coop.txt).  Returns as mul_base_ct.
// LANES is a launch argument (public, wavefront-uniform; a power of two, 2 .. 64): one kernel per curve serves 64 / 16 / 4.
template <int N>
__device__ __forceinline__ uint32_t mul_base_coop(feT<N> &x, feT<N> &y, const uint8_t *__restrict__ kbytes,
                                                  const uint32_t *__restrict__ gtab8, const int LANES)
{
    const int lane = (int)(threadIdx.x & (unsigned)(LANES - 1));
    feT<N> b;
#pragma unroll
    for (int i = 0; i < N; ++i) b.v[i] = curve_b<N>()[i];
    projT<N> acc;
    fe_set_zero(acc.X); fe_set_one(acc.Y); fe_set_zero(acc.Z);          // O
#pragma unroll 1
    for (int t = 0; t < ((8 * N + LANES - 1) >> __builtin_ctz((unsigned)LANES)); ++t) {
        const int w = lane + LANES * t;
        const bool mine = w < 8 * N;                 // not a secret
        const int wc = mine ? w : 0;
        uint32_t dg = ((uint32_t)kbytes[wc >> 1] >> (4 * (wc & 1))) & 15u;
        dg = mine ? dg : 0u;
        const uint32_t *row = gtab8 + ((size_t)(wc >> 1) * GT8_ENTRIES) * (2 * N);
        const int sh = 4 * (wc & 1);
        affT<N> E;
        fe_set_zero(E.x); fe_set_zero(E.y);
#pragma unroll 1
        for (int j = 1; j < 16; ++j) {
            const uint32_t m = ct_eq_small(dg, (uint32_t)j);
            const uint4 *e = reinterpret_cast<const uint4 *>(row + (size_t)(j << sh) * (2 * N));
#pragma unroll
            for (int l = 0; l < N / 4; ++l) {
                const uint4 ex = e[l], ey = e[N / 4 + l];
                E.x.v[4 * l + 0] = bitop3<0xF8>(E.x.v[4 * l + 0], ex.x, m);
                E.x.v[4 * l + 1] = bitop3<0xF8>(E.x.v[4 * l + 1], ex.y, m);
                E.x.v[4 * l + 2] = bitop3<0xF8>(E.x.v[4 * l + 2], ex.z, m);
                E.x.v[4 * l + 3] = bitop3<0xF8>(E.x.v[4 * l + 3], ex.w, m);
                E.y.v[4 * l + 0] = bitop3<0xF8>(E.y.v[4 * l + 0], ey.x, m);
                E.y.v[4 * l + 1] = bitop3<0xF8>(E.y.v[4 * l + 1], ey.y, m);
                E.y.v[4 * l + 2] = bitop3<0xF8>(E.y.v[4 * l + 2], ey.z, m);
                E.y.v[4 * l + 3] = bitop3<0xF8>(E.y.v[4 * l + 3], ey.w, m);
            }
        }
        projT<N> sum = acc;
        proj_madd_complete(sum, E, b);               // digit 0: E = (0, 0) is not a point; the old accumulator is kept
        const uint32_t keep = ct_eq_small(dg, 0u);
#pragma unroll
        for (int l = 0; l < N; ++l) {
            acc.X.v[l] = ct_sel(keep, acc.X.v[l], sum.X.v[l]);
            acc.Y.v[l] = ct_sel(keep, acc.Y.v[l], sum.Y.v[l]);
            acc.Z.v[l] = ct_sel(keep, acc.Z.v[l], sum.Z.v[l]);
        }
    }
#pragma unroll 1
    for (int s = 1; s < LANES; s <<= 1) {              // partners stay inside the aligned group of LANES lanes
        projT<N> o;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            o.X.v[l] = (uint32_t)__shfl_xor((int)acc.X.v[l], s, 64);
            o.Y.v[l] = (uint32_t)__shfl_xor((int)acc.Y.v[l], s, 64);
            o.Z.v[l] = (uint32_t)__shfl_xor((int)acc.Z.v[l], s, 64);
        }
        proj_add_complete(acc, o, b);
    }
    return proj_to_affine_ct(x, y, acc);
}

// LANES lanes per scalar (block = one wavefront = 64 / LANES scalars); modes and outputs as bign_mulbase_ct_kernel.
// MODE, X_ONLY and LANES are launch arguments (public, wavefront-uniform): one kernel per curve in the product library.
template <int N>
__global__ __launch_bounds__(64)
void bign_mulbase_coop_kernel(const uint8_t *__restrict__ scalars, size_t n, uint32_t *__restrict__ codes,
                              uint8_t *__restrict__ xy_out, const uint32_t *__restrict__ gtab8, const int MODE, const int X_ONLY,
                              const int LANES)
{
    constexpr int NO = 4 * N;
    const int lsh = __builtin_ctz((unsigned)LANES);
    const size_t idx = (size_t)blockIdx.x * (64u >> lsh) + (threadIdx.x >> lsh);
    if (idx >= n) return;                            // a whole group leaves: no partner of a working lane goes missing
    uint32_t valid = ~0u;
    if (MODE == 1) {
        uint32_t k[N];
        load_words_bytes(k, scalars + NO * idx);
        valid = ct_in_range_q(k);
    }
    feT<N> x, y;
    const uint32_t inf = mul_base_coop<N>(x, y, scalars + NO * idx, gtab8, LANES);
    if (MODE == 2) valid = ~inf;
    if ((threadIdx.x & (unsigned)(LANES - 1)) != 0) return;
    if (MODE == 1) codes[idx] = ct_sel(valid, (uint32_t)ERR_OK, ERR_BAD_PRIVKEY_V);
    if (MODE == 2) codes[idx] = ct_sel(inf, (uint32_t)ERR_BAD_PARAMS, (uint32_t)ERR_OK);
    uint32_t *o = reinterpret_cast<uint32_t *>(xy_out + (size_t)(X_ONLY ? NO : 2 * NO) * idx);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = x.v[i] & valid;
    if (!X_ONLY) {
#pragma unroll
        for (int i = 0; i < N; ++i) o[N + i] = y.v[i] & valid;
    }
}

// scalars: n x 4N octets (LE).
// MODE 1 (bignPubkeyCalc): the scalars are private keys d -- codes[i] = ERR_OK / ERR_BAD_PRIVKEY
//         (bign_misc.c:399-403); an invalid key leaves an all-zero public key behind.
// MODE 2 (bignKeypairGen): any d below 2^(2l) is multiplied, as the reference does with its draw below p
//         (bign_misc.c:209-218); codes[i] = ERR_BAD_PARAMS when d G = O (d = q), else ERR_OK.
// MODE 0 (signing): no codes; every lane computes.
// xy_out: n x 8N octets (x || y), or with X_ONLY n x 4N octets.
#ifndef SIGN_MULBASE_WAVES
#define SIGN_MULBASE_WAVES 3        // 168 VGPRs, 15 spilled: +2.2 % over 2 (186 VGPRs); 4 (128 VGPRs, 110-124 spilled): -31 %
#endif
template <int N, int FORM = 0>          // FORM 0: 4-bit windows, 1: signed 6-bit + complete additions, 2: + Jacobian
__global__ __launch_bounds__(256, (N == 8 ? SIGN_MULBASE_WAVES : 1))
void bign_mulbase_ct_kernel(const uint8_t *__restrict__ scalars, size_t n, uint32_t *__restrict__ codes,
                            uint8_t *__restrict__ xy_out, const uint32_t *__restrict__ gtab8,      // FORM > 0: the signed 6-bit table
                            const int MODE, const int X_ONLY)                                      // launch arguments: public, uniform
{
    constexpr int NO = 4 * N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t k[N];
    load_words_bytes(k, scalars + NO * idx);
    uint32_t valid = ~0u;
    if (MODE == 1) {
        valid = ct_in_range_q(k);
        codes[idx] = ct_sel(valid, (uint32_t)ERR_OK, ERR_BAD_PRIVKEY_V);
    }
    feT<N> x, y;
    uint32_t inf;
    if constexpr (FORM == 2) inf = mul_base_ct6<N, true>(x, y, k, gtab8);
    else if constexpr (FORM == 1) inf = mul_base_ct6<N, false>(x, y, k, gtab8);
    else inf = mul_base_ct(x, y, k, gtab8);
    if (MODE == 2) {
        valid = ~inf;
        codes[idx] = ct_sel(inf, (uint32_t)ERR_BAD_PARAMS, (uint32_t)ERR_OK);
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(xy_out + (size_t)(X_ONLY ? NO : 2 * NO) * idx);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = x.v[i] & valid;
    if (!X_ONLY) {
#pragma unroll
        for (int i = 0; i < N; ++i) o[N + i] = y.v[i] & valid;
    }
}

// (round 6) The same window walk with the accumulator on nine signed 29-bit limbs (bign_fe29.hpp): the Jacobian mixed addition is
// 8 multiplications + 3 squarings, each ONE asm block of 135 / 99 instructions with no carry flags, against ~190 / ~150 VALU
// instructions (a third of them v_addc_co and v_mov) of the 32-bit product-scanning code -- per window 9 400 -> ~6 900 VALU cycles
// (profiles/r06_sign_l29_ab.txt).  Constant-time as before: the limb arithmetic is masks, shifts and multiply-adds, the digit's
// entry is looked up in the bank-private LDS copies, sign and "keep" are applied by ct_sel.  The table stays in 32-bit words
// (converted on load: 2 instructions per limb); the result goes back to words for the fixed-count inversion.
template <int WB, int WGL>
__device__ __forceinline__ uint32_t mul_base_ct_lds16_l29(feT<8> &x, feT<8> &y, const uint32_t (&k)[8], const uint4 *__restrict__ tabw,
                                                          uint4 *s_row, const int x_only)
{
    constexpr int N = 8, L = LZ<N>::L;
    constexpr int W = WinW<N, WB>::W, ENT = WinW<N, WB>::ENT, OW = ENT * N / 2;     // 16-octet words of a row
    static_assert(OW * 16 % WGL == 0 && WGL % 16 == 0, "the refill covers the row with whole passes of the workgroup");
    const unsigned tid = threadIdx.x, rep = tid & 15u;
    uint32_t kk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) kk[i] = k[i];
    jac29 J;
#pragma unroll
    for (int i = 0; i < L; ++i) { J.X.l[i] = 0; J.Y.l[i] = i == 0; J.Z.l[i] = 0; }
    uint32_t at_inf = ~0u;
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        __syncthreads();
        {
            const uint4 *row = tabw + (size_t)w * OW;
#pragma unroll
            for (int i = 0; i < OW * 16 / WGL; ++i) {
                const unsigned q = (tid >> 4) + (unsigned)(WGL / 16) * i;   // the 16 lanes of a copy-group fetch the same word
                s_row[q * 16u + rep] = row[q];
            }
        }
        __syncthreads();
        const uint32_t t = (kk[0] & (uint32_t)(2 * ENT - 1)) + carry;   // 0 .. 2 ENT
#pragma unroll
        for (int i = 0; i < N - 1; ++i) kk[i] = __builtin_amdgcn_alignbit(kk[i + 1], kk[i], WB);
        kk[N - 1] >>= WB;
        carry = (t + (uint32_t)ENT) >> WB;
        const uint32_t d = t - (carry << WB);
        const uint32_t neg = (uint32_t)((int32_t)d >> 31);
        const uint32_t mag = (d ^ neg) - neg;                           // 0 .. ENT
        const uint32_t e = (mag - 1u) & (uint32_t)(ENT - 1);            // digit 0 reads entry ENT: looked up, added, not used
        affT<N> E;
#pragma unroll
        for (int l = 0; l < N / 4; ++l) {
            const uint4 vx = s_row[(e * (N / 2) + l) * 16u + rep], vy = s_row[(e * (N / 2) + N / 4 + l) * 16u + rep];
            E.x.v[4 * l] = vx.x; E.x.v[4 * l + 1] = vx.y; E.x.v[4 * l + 2] = vx.z; E.x.v[4 * l + 3] = vx.w;
            E.y.v[4 * l] = vy.x; E.y.v[4 * l + 1] = vy.y; E.y.v[4 * l + 2] = vy.z; E.y.v[4 * l + 3] = vy.w;
        }
        aff29 E29;
        f29_from_words(E29.x, E.x);
        f29_from_words(E29.y, E.y);
#pragma unroll
        for (int l = 0; l < L; ++l) E29.y.l[l] = (int32_t)ct_sel(neg, (uint32_t)(-E29.y.l[l]), (uint32_t)E29.y.l[l]);
        const uint32_t keep = ct_eq_small(mag, 0u);
        jac29 sum = J;
        jac29_madd(sum, E29);                                           // digit 0 or accumulator still O: computed, not used
        const uint32_t set = at_inf & ~keep;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            J.X.l[l] = (int32_t)ct_sel(keep, (uint32_t)J.X.l[l], ct_sel(set, (uint32_t)E29.x.l[l], (uint32_t)sum.X.l[l]));
            J.Y.l[l] = (int32_t)ct_sel(keep, (uint32_t)J.Y.l[l], ct_sel(set, (uint32_t)E29.y.l[l], (uint32_t)sum.Y.l[l]));
            J.Z.l[l] = (int32_t)ct_sel(keep, (uint32_t)J.Z.l[l], ct_sel(set, l == 0 ? 1u : 0u, (uint32_t)sum.Z.l[l]));
        }
        at_inf &= keep;
    }
    jacT<N> Jw;
    f29_to_words(Jw.X, J.X);
    f29_to_words(Jw.Y, J.Y);
    f29_to_words(Jw.Z, J.Z);
#pragma unroll
    for (int l = 0; l < N; ++l) Jw.Z.v[l] &= ~at_inf;
    feT<N> zc;
    fe_canon(zc, Jw.Z);
    // ---- 1 / Z, SHARED inside the workgroup (round 6).  The fixed-count division steps are ~20 000 instructions; with every lane
    // inverting its own Z that was a quarter of this kernel on all 16 wavefronts of the workgroup.  Montgomery's trick over groups of
    // INV_K = 4 lanes' values: Z (zero replaced by one) goes to LDS -- the window rows are dead by now --, the first WGL / 4 lanes (one
    // wavefront per SIMD) each take four values: prefix products, ONE inversion, back-substitution (9 multiplications), and every lane
    // reads its own inverse back.  Who does what depends on the lane index only (public); the staged values sit at lane-indexed
    // addresses and are wiped afterwards.  A lane whose Z was zero gets an inverse that fails the z zi = 1 check below, as before.
    constexpr int INV_K = 4;
    static_assert(WGL % (64 * INV_K) == 0, "whole wavefronts of inverting lanes");
    uint32_t *s_z = reinterpret_cast<uint32_t *>(s_row);                     // [word][lane]: N x WGL words; prefix products behind it
    uint32_t *s_p = s_z + N * WGL;                                           // [k = 1 .. INV_K-2][word][inverting lane]
    const uint32_t z_is0 = ct_is_zero(zc.v);
    __syncthreads();                                                        // every lane is through with the window rows
#pragma unroll
    for (int l = 0; l < N; ++l) s_z[l * WGL + tid] = ct_sel(z_is0, l == 0 ? 1u : 0u, zc.v[l]);
    __syncthreads();
    if (tid < WGL / INV_K) {
        constexpr unsigned G = WGL / INV_K;                                  // value j of lane t = the Z of lane j G + t
        feT<N> zj, acc, t0;
#pragma unroll
        for (int l = 0; l < N; ++l) acc.v[l] = s_z[l * WGL + tid];
#pragma unroll 1
        for (int j = 1; j < INV_K; ++j) {
            if (j > 1) {
#pragma unroll
                for (int l = 0; l < N; ++l) s_p[((j - 2) * N + l) * G + tid] = acc.v[l];     // product of values 0 .. j-1
            }
#pragma unroll
            for (int l = 0; l < N; ++l) zj.v[l] = s_z[l * WGL + j * G + tid];
            fe_mul(acc, acc, zj);
        }
        fe_canon(acc, acc);
        feT<N> inv = fe_inv_safegcd<N, true>(acc);                           // 1 / (z_0 z_1 z_2 z_3)
#pragma unroll 1
        for (int j = INV_K - 1; j >= 1; --j) {
#pragma unroll
            for (int l = 0; l < N; ++l) t0.v[l] = j > 1 ? s_p[((j - 2) * N + l) * G + tid] : s_z[l * WGL + tid];   // product before value j
#pragma unroll
            for (int l = 0; l < N; ++l) zj.v[l] = s_z[l * WGL + j * G + tid];
            fe_mul(t0, t0, inv);                                             // 1 / z_j
            fe_mul(inv, inv, zj);
#pragma unroll
            for (int l = 0; l < N; ++l) s_z[l * WGL + j * G + tid] = t0.v[l];
        }
#pragma unroll
        for (int l = 0; l < N; ++l) s_z[l * WGL + tid] = inv.v[l];           // 1 / z_0
#pragma unroll 1
        for (int j = 0; j < (INV_K - 2) * N; ++j) s_p[j * G + tid] = 0;      // the prefix products leave LDS
    }
    __syncthreads();
    feT<N> zi;
#pragma unroll
    for (int l = 0; l < N; ++l) { zi.v[l] = s_z[l * WGL + tid]; s_z[l * WGL + tid] = 0; }
    feT<N> chk, zi2;
    fe_mul(chk, zc, zi);
    fe_canon(chk, chk);
    chk.v[0] ^= 1u;
    fe_sqr(zi2, zi);
    fe_mul(x, Jw.X, zi2);
    fe_canon(x, x);
    fe_set_zero(y);
    if (!x_only) {
        fe_mul(zi2, zi2, zi);
        fe_mul(y, Jw.Y, zi2);
        fe_canon(y, y);
    }
    const uint32_t inf = ct_is_zero(zc.v) | ~ct_is_zero(chk.v);
#pragma unroll
    for (int l = 0; l < N; ++l) { x.v[l] &= ~inf; y.v[l] &= ~inf; }
    return inf;
}

// Signed 8-bit windows (33 additions on the 256-bit curve): 128 entries x 64 octets = 8 KiB per row, which fits LDS in SIXTEEN
// copies of 16-octet words, copy r at byte address (word * 16 + r) * 16, read with ds_read_b128 by lane l from copy l mod 16.
// ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS table) -- and within each of them l mod 16 takes every value once (also within plain runs of 16
// lanes), so the 16 lanes of a group read 16 disjoint 16-octet slots = all 64 banks once, whatever their entries: no
// conflict, SQ_LDS_BANK_CONFLICT = 0 for every key class (profiles/r04_sign_lds.txt).  Four reads per look-up.
// WGL = lanes of the workgroup (1024, or 512 for batches that would leave CUs empty otherwise: two wavefronts per SIMD still reach
// ~95 % of the multiply-add rate).
template <int N, int WB, int WGL>
__device__ __forceinline__ uint32_t mul_base_ct_lds16(feT<N> &x, feT<N> &y, const uint32_t (&k)[N], const uint4 *__restrict__ tabw,
                                                      uint4 *s_row, const int x_only)
{
    constexpr int W = WinW<N, WB>::W, ENT = WinW<N, WB>::ENT, OW = ENT * N / 2;     // 16-octet words of a row
    static_assert(OW * 16 % WGL == 0 && N % 4 == 0 && WGL % 16 == 0, "the refill covers the row with whole passes of the workgroup");
    const unsigned tid = threadIdx.x, rep = tid & 15u;
    uint32_t kk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) kk[i] = k[i];
    jacT<N> J;
    fe_set_zero(J.X); fe_set_one(J.Y); fe_set_zero(J.Z);
    uint32_t at_inf = ~0u;
    feT<N> one;
    fe_set_one(one);
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        __syncthreads();
        {
            const uint4 *row = tabw + (size_t)w * OW;
#pragma unroll
            for (int i = 0; i < OW * 16 / WGL; ++i) {
                const unsigned q = (tid >> 4) + (unsigned)(WGL / 16) * i;   // the 16 lanes of a copy-group fetch the same word
                s_row[q * 16u + rep] = row[q];
            }
        }
        __syncthreads();
        const uint32_t t = (kk[0] & (uint32_t)(2 * ENT - 1)) + carry;   // 0 .. 2 ENT
#pragma unroll
        for (int i = 0; i < N - 1; ++i) kk[i] = __builtin_amdgcn_alignbit(kk[i + 1], kk[i], WB);
        kk[N - 1] >>= WB;
        carry = (t + (uint32_t)ENT) >> WB;
        const uint32_t d = t - (carry << WB);
        const uint32_t neg = (uint32_t)((int32_t)d >> 31);
        const uint32_t mag = (d ^ neg) - neg;                           // 0 .. ENT
        const uint32_t e = (mag - 1u) & (uint32_t)(ENT - 1);            // digit 0 reads entry ENT: looked up, added, not used
        affT<N> E;
#pragma unroll
        for (int l = 0; l < N / 4; ++l) {
            const uint4 vx = s_row[(e * (N / 2) + l) * 16u + rep], vy = s_row[(e * (N / 2) + N / 4 + l) * 16u + rep];
            E.x.v[4 * l] = vx.x; E.x.v[4 * l + 1] = vx.y; E.x.v[4 * l + 2] = vx.z; E.x.v[4 * l + 3] = vx.w;
            E.y.v[4 * l] = vy.x; E.y.v[4 * l + 1] = vy.y; E.y.v[4 * l + 2] = vy.z; E.y.v[4 * l + 3] = vy.w;
        }
        feT<N> ny;
        fe_neg(ny, E.y);
#pragma unroll
        for (int l = 0; l < N; ++l) E.y.v[l] = ct_sel(neg, ny.v[l], E.y.v[l]);
        const uint32_t keep = ct_eq_small(mag, 0u);
        jacT<N> sum = J;
        jac_madd_ct(sum, E);
        const uint32_t set = at_inf & ~keep;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            J.X.v[l] = ct_sel(keep, J.X.v[l], ct_sel(set, E.x.v[l], sum.X.v[l]));
            J.Y.v[l] = ct_sel(keep, J.Y.v[l], ct_sel(set, E.y.v[l], sum.Y.v[l]));
            J.Z.v[l] = ct_sel(keep, J.Z.v[l], ct_sel(set, one.v[l], sum.Z.v[l]));
        }
        at_inf &= keep;
    }
#pragma unroll
    for (int l = 0; l < N; ++l) J.Z.v[l] &= ~at_inf;
    feT<N> zc;
    fe_canon(zc, J.Z);
    const feT<N> zi = fe_inv_safegcd<N, true>(zc);
    feT<N> chk, zi2;
    fe_mul(chk, zc, zi);
    fe_canon(chk, chk);
    chk.v[0] ^= 1u;
    fe_sqr(zi2, zi);
    fe_mul(x, J.X, zi2);
    fe_canon(x, x);
    fe_set_zero(y);
    if (!x_only) {
        fe_mul(zi2, zi2, zi);
        fe_mul(y, J.Y, zi2);
        fe_canon(y, y);
    }
    const uint32_t inf = ct_is_zero(zc.v) | ~ct_is_zero(chk.v);
#pragma unroll
    for (int l = 0; l < N; ++l) { x.v[l] &= ~inf; y.v[l] &= ~inf; }
    return inf;
}

// The same walk with the accumulator in "XYZZ" coordinates (X, Y, ZZ = Z^2, ZZZ = Z^3; x = X / ZZ, y = Y / ZZZ): the mixed addition
// is 8M + 2S instead of 8M + 3S (no Z^2 to recompute) for one more coordinate to carry -- U2 = x2 ZZ1, S2 = y2 ZZZ1, P = U2 - X1,
// R = S2 - Y1, PP = P^2, PPP = P PP, Q = X1 PP, X3 = R^2 - PPP - 2Q, Y3 = R (Q - X3) - Y1 PPP, ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP
// (the same group operation with the same exceptional cases as jac_madd_ct: P = 0, excluded by the schedule).  To keep the
// working set inside 128 VGPRs the selected entry is NOT held across the addition: the row is still in LDS when the new
// coordinates are blended in, so the "first non-zero digit" case reads it again (8 more ds_read_b64).
template <int N, int WB>
__device__ __forceinline__ uint32_t mul_base_ct_lds_xyzz(feT<N> &x, feT<N> &y, const uint32_t (&k)[N], const uint64_t *__restrict__ tabw,
                                                         uint64_t *s_row, const int x_only)
{
    constexpr int W = WinW<N, WB>::W, ENT = WinW<N, WB>::ENT, QW = ENT * N;
    static_assert(QW * 32 % 1024 == 0, "the refill is written for 1024 lanes");
    const unsigned tid = threadIdx.x, rep = tid & 31u;
    uint32_t kk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) kk[i] = k[i];
    feT<N> X, Y, ZZ, ZZZ;
    fe_set_zero(X); fe_set_one(Y); fe_set_zero(ZZ); fe_set_zero(ZZZ);
    uint32_t at_inf = ~0u, carry = 0;
    const auto read_entry = [&](affT<N> &E, uint32_t e, uint32_t neg) {
#pragma unroll
        for (int l = 0; l < N / 2; ++l) {
            const uint64_t vx = s_row[(e * N + l) * 32u + rep], vy = s_row[(e * N + N / 2 + l) * 32u + rep];
            E.x.v[2 * l] = (uint32_t)vx; E.x.v[2 * l + 1] = (uint32_t)(vx >> 32);
            E.y.v[2 * l] = (uint32_t)vy; E.y.v[2 * l + 1] = (uint32_t)(vy >> 32);
        }
        feT<N> ny;
        fe_neg(ny, E.y);
#pragma unroll
        for (int l = 0; l < N; ++l) E.y.v[l] = ct_sel(neg, ny.v[l], E.y.v[l]);
    };
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        __syncthreads();
        {
            const uint64_t *row = tabw + (size_t)w * QW;
#pragma unroll
            for (int i = 0; i < QW * 32 / 1024; ++i) {
                const unsigned q = (tid >> 5) + 32u * i;
                s_row[q * 32u + rep] = row[q];
            }
        }
        __syncthreads();
        const uint32_t t = (kk[0] & (uint32_t)(2 * ENT - 1)) + carry;
#pragma unroll
        for (int i = 0; i < N - 1; ++i) kk[i] = __builtin_amdgcn_alignbit(kk[i + 1], kk[i], WB);
        kk[N - 1] >>= WB;
        carry = (t + (uint32_t)ENT) >> WB;
        const uint32_t d = t - (carry << WB);
        const uint32_t neg = (uint32_t)((int32_t)d >> 31);
        const uint32_t mag = (d ^ neg) - neg;
        const uint32_t e = (mag - 1u) & (uint32_t)(ENT - 1);
        const uint32_t keep = ct_eq_small(mag, 0u);
        const uint32_t set = at_inf & ~keep;
        feT<N> nX, nY, nZZ, nZZZ;
        {
            feT<N> P, R, PP, PPP, Q, t0, t1;
            {
                affT<N> E;
                read_entry(E, e, neg);
                fe_mul(P, E.x, ZZ);                                     // U2
                fe_mul(R, E.y, ZZZ);                                    // S2
            }
            fe_sub(P, P, X);
            fe_sub(R, R, Y);
            fe_sqr(PP, P);
            fe_mul(PPP, P, PP);
            fe_mul(Q, X, PP);
            fe_mul(nZZ, ZZ, PP);
            fe_mul(nZZZ, ZZZ, PPP);
            fe_sqr(t0, R);
            fe_sub(t0, t0, PPP);
            fe_dbl(t1, Q);
            fe_sub(nX, t0, t1);
            fe_sub(t0, Q, nX);
            fe_mul(t0, R, t0);
            fe_mul(t1, Y, PPP);
            fe_sub(nY, t0, t1);
        }
        {
            affT<N> E;                                                  // again, for the lanes whose accumulator starts here
            read_entry(E, e, neg);
#pragma unroll
            for (int l = 0; l < N; ++l) {
                const uint32_t o = l == 0 ? 1u : 0u;
                X.v[l] = ct_sel(keep, X.v[l], ct_sel(set, E.x.v[l], nX.v[l]));
                Y.v[l] = ct_sel(keep, Y.v[l], ct_sel(set, E.y.v[l], nY.v[l]));
                ZZ.v[l] = ct_sel(keep, ZZ.v[l], ct_sel(set, o, nZZ.v[l]));
                ZZZ.v[l] = ct_sel(keep, ZZZ.v[l], ct_sel(set, o, nZZZ.v[l]));
            }
        }
        at_inf &= keep;
    }
#pragma unroll
    for (int l = 0; l < N; ++l) ZZ.v[l] &= ~at_inf;
    feT<N> zc;
    fe_canon(zc, ZZ);
    const feT<N> zi = fe_inv_safegcd<N, true>(zc);
    feT<N> chk;
    fe_mul(chk, zc, zi);
    fe_canon(chk, chk);
    chk.v[0] ^= 1u;                                                     // 0 iff ZZ * ZZ^-1 == 1
    fe_mul(x, X, zi);
    fe_canon(x, x);
    fe_set_zero(y);
    if (!x_only) {                                                      // 1 / ZZZ = ZZZ / ZZ^3 (ZZZ^2 = ZZ^3)
        feT<N> z2, z3;
        fe_sqr(z2, zi);
        fe_mul(z3, z2, zi);
        fe_mul(y, Y, ZZZ);
        fe_mul(y, y, z3);
        fe_canon(y, y);
    }
    const uint32_t inf = ct_is_zero(zc.v) | ~ct_is_zero(chk.v);
#pragma unroll
    for (int l = 0; l < N; ++l) { x.v[l] &= ~inf; y.v[l] &= ~inf; }
    return inf;
}

// the LDS look-up form: modes and outputs as bign_mulbase_ct_kernel; blocks of 1024 lanes, all of which walk the windows
// (a lane beyond n multiplies by 0 and writes nothing: the barriers need every lane)
template <int N, int WB, bool XYZZ, int WGL = 1024, bool L29 = (N == 8 && WB == 8)>
__global__ __launch_bounds__(WGL)
void bign_mulbase_lds_kernel(const uint8_t *__restrict__ scalars, size_t n, uint32_t *__restrict__ codes,
                             uint8_t *__restrict__ xy_out, const uint64_t *__restrict__ tabw, const int MODE, const int X_ONLY)
{
    constexpr int NO = 4 * N;
    extern __shared__ __attribute__((aligned(16))) uint64_t s_row_dyn[];
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < n;                       // public
    uint32_t k[N];
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = 0;
    if (live) load_words_bytes(k, scalars + NO * idx);
    uint32_t valid = ~0u;
    if (MODE == 1) valid = ct_in_range_q(k);
    feT<N> x, y;
    uint32_t inf;
    if constexpr (WB == 8 && L29)
        inf = mul_base_ct_lds16_l29<WB, WGL>(x, y, k, reinterpret_cast<const uint4 *>(tabw), reinterpret_cast<uint4 *>(s_row_dyn), X_ONLY);
    else if constexpr (WB == 8)
        inf = mul_base_ct_lds16<N, WB, WGL>(x, y, k, reinterpret_cast<const uint4 *>(tabw), reinterpret_cast<uint4 *>(s_row_dyn), X_ONLY);
    else if constexpr (XYZZ) inf = mul_base_ct_lds_xyzz<N, WB>(x, y, k, tabw, s_row_dyn, X_ONLY);
    else inf = mul_base_ct_lds<N, WB>(x, y, k, tabw, s_row_dyn, X_ONLY);
    if (!live) return;
    if (MODE == 1) codes[idx] = ct_sel(valid, (uint32_t)ERR_OK, ERR_BAD_PRIVKEY_V);
    if (MODE == 2) {
        valid = ~inf;
        codes[idx] = ct_sel(inf, (uint32_t)ERR_BAD_PARAMS, (uint32_t)ERR_OK);
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(xy_out + (size_t)(X_ONLY ? NO : 2 * NO) * idx);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = x.v[i] & valid;
    if (!X_ONLY) {
#pragma unroll
        for (int i = 0; i < N; ++i) o[N + i] = y.v[i] & valid;
    }
}

// ------------------------------------------------ belt-hash of a short message ---
// words of the message live in an LDS row per lane (secret data: the row is lane-private); nw = number of
// 32-bit words that hold message bytes, len = message length in octets (wavefront-uniform).
// The hash starts from the standard's initial value, or from the state the OID's leading whole blocks left (OidArg,
// bign_kernels.hip): len counts the octets in the row, oid.pre_len those absorbed before it.
template <class Tab>
__device__ __forceinline__ void belt_hash_row(const Tab &T, uint32_t (&h)[8], const uint32_t *row, uint32_t len, const OidArg &oid)
{
    uint32_t s[4], X[8], s1[4];
    oid_hash_start(h, s, oid);
    const uint32_t nblk = (len + 31) / 32;
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; ++b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) X[i] = row[8 * b + i];          // the row is zero-padded to whole blocks
        belt_compress(T, s1, h, X);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] ^= s1[i];
    }
    X[0] = (len + oid.pre_len) << 3; X[1] = (len + oid.pre_len) >> 29; X[2] = 0; X[3] = 0;   // <bit length>_128 || s (belt_hash.c:120-135)
    X[4] = s[0]; X[5] = s[1]; X[6] = s[2]; X[7] = s[3];
    belt_compress(T, s1, h, X);
}

// message = oid || a || b assembled into the lane's row: a_words words of a, b_len octets of b (b may be null).
// The OID is wavefront-uniform; data words are funnel-shifted by the OID length mod 4 (as bign_tail_kernel does).
__device__ __forceinline__ void row_put_bytes(uint32_t *row, uint32_t pos, uint32_t byte)
{
    row[pos >> 2] |= byte << (8 * (pos & 3u));
}

constexpr int SIGN_WG = 256;
constexpr int SIGN_T_MAX = 64;                       // longest additional input t (octets) the device path takes

// algorithm 6.3.3 (bign_sign.c:192-216): theta = belt-hash(oid || d || t), k = H, repeat k <- belt-wbl_theta(k) until
// 0 < k < q.  Also range-checks d (:185-189).  Writes k (4N octets per signature) and status.
// t: n x t_len octets (t_stride = t_len) or one shared string (t_stride = 0); may be null with t_len = 0.
// row_words: 32-bit words of a lane's LDS row (odd; sign_row_words() of the message this launch hashes).  The rows used to be
// sized for the longest OID and t the kernel accepts (65 words: with the 64 KiB table one workgroup of 256 lanes per CU, i.e.
// ONE wavefront per SIMD walking dependent block encryptions); sized for the message at hand, 1024 lanes share a table.
// (workgroup size: a launch parameter, 256 .. 1024 -- one kernel per curve)
template <int N>
__global__ __launch_bounds__(1024)
void bign_sign_nonce_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ privkeys,
                            const uint8_t *__restrict__ t, uint32_t t_len, uint32_t t_stride,
                            const uint8_t *__restrict__ theta_in, size_t n, OidArg oid, QArg<N> qa,
                            uint32_t *__restrict__ status, uint8_t *__restrict__ k_out, uint32_t row_words)
{
    constexpr int NO = 4 * N;
    const uint32_t ROW = row_words;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint8_t *s_tab = s_dyn;
    uint32_t *s_rows = reinterpret_cast<uint32_t *>(s_dyn + BeltTabTwo::kBytes);
    BeltTabTwo::fill(s_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const BeltTabTwoP T(s_tab);         // (round 4) the SDWA-address form of the same bank-private table: 8 instead of 12 VALU per G-box
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;

    uint32_t d[N];
    load_words_bytes(d, privkeys + NO * idx);
    const uint32_t d_ok = ct_in_range_q(d, qa);
    status[idx] = ct_sel(d_ok, ST_PENDING, ERR_BAD_PRIVKEY_V);

    // ---- theta = belt-hash(oid || d || t) (or taken from the caller: additional input longer than SIGN_T_MAX is
    // hashed by ONE ragged belt-hash launch over oid || d || t on the bank-private table, capi_bign.hip sign_batch_host)
    uint32_t theta[8];
    if (theta_in) {
#pragma unroll
        for (int i = 0; i < 8; ++i) theta[i] = reinterpret_cast<const uint32_t *>(theta_in + 32 * idx)[i];
    } else {
    uint32_t *row = s_rows + threadIdx.x * ROW;
    const uint32_t len = oid.len + NO + t_len;
    const uint32_t nwords = (len + 31) / 32 * 8;
    for (uint32_t i = 0; i < nwords; ++i) row[i] = 0;
    for (uint32_t i = 0; i < oid.len; ++i) row_put_bytes(row, i, oid.der[i]);
    {
        const uint32_t sh = (oid.len & 3u) * 8u, w0 = oid.len >> 2;          // uniform
#pragma unroll
        for (int i = 0; i < N; ++i) {
            row[w0 + i] |= d[i] << sh;
            if (sh) row[w0 + i + 1] |= d[i] >> (32 - sh);
        }
    }
    if (t_len) {
        const uint8_t *tp = t + (size_t)t_stride * idx;
        for (uint32_t i = 0; i < t_len; ++i) row_put_bytes(row, oid.len + NO + i, tp[i]);
    }
    belt_hash_row(T, theta, row, len, oid);
    for (uint32_t i = 0; i < nwords; ++i) row[i] = 0;                        // wipe d from LDS
    }

    // ---- k = belt-wbl_theta(H), repeated until 0 < k < q.  belt-wbl on NB = N / 4 blocks (belt_wbl.c:58-152):
    // 2 NB rounds  s = r_1 ^ .. ^ r_{NB-1};  (r_1 .. r_NB) <- (r_2, .., r_{NB-1}, r_NB ^ E(s) ^ <i>, s)
    constexpr int NB = N / 4;
    uint32_t r[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const uint4 a = *reinterpret_cast<const uint4 *>(hashes + NO * idx + 16 * j);
        r[j][0] = a.x; r[j][1] = a.y; r[j][2] = a.z; r[j][3] = a.w;
    }
    uint32_t kk[N];
    uint32_t done = 0;
#pragma unroll 1
    for (;;) {
        uint32_t c[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[j][i] = r[j][i];
#pragma unroll
        for (int round = 1; round <= 2 * NB; ++round) {
            uint32_t s[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] = c[0][i];
#pragma unroll
                for (int j = 1; j < NB - 1; ++j) s[i] ^= c[j][i];
            }
            uint32_t e[4] = {s[0], s[1], s[2], s[3]};
            belt_encr(T, e, theta);
            e[0] ^= (uint32_t)round;
            uint32_t last[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) last[i] = c[NB - 1][i] ^ e[i];
#pragma unroll
            for (int j = 0; j + 2 < NB; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[j][i] = c[j + 1][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[NB - 2][i] = last[i]; c[NB - 1][i] = s[i]; }
        }
        // lanes that already have their k keep it; the others take the new value
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) r[j][i] = ct_sel(done, r[j][i], c[j][i]);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) kk[4 * j + i] = r[j][i];
        done |= ct_in_range_q(kk, qa);
        if (__all(done != 0)) break;          // fails with probability < 2^-126 per signature (see the header; a non-standard
                                              // q may sit anywhere in [2^(2l-1), 2^2l): up to one rejection in two there)
    }
    uint32_t *ko = reinterpret_cast<uint32_t *>(k_out + NO * idx);
#pragma unroll
    for (int i = 0; i < N; ++i) ko[i] = kk[i];
}

// bignSign with the one-time key supplied (bign_sign.c:71-82 after the rng): d and k range checks only
template <int N>
__global__ __launch_bounds__(256)
void bign_sign_kcheck_kernel(const uint8_t *__restrict__ privkeys, const uint8_t *__restrict__ ks, size_t n, QArg<N> qa,
                             uint32_t *__restrict__ status)
{
    constexpr int NO = 4 * N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t d[N], k[N];
    load_words_bytes(d, privkeys + NO * idx);
    load_words_bytes(k, ks + NO * idx);
    const uint32_t d_ok = ct_in_range_q(d, qa), k_ok = ct_in_range_q(k, qa);
    status[idx] = ct_sel(d_ok, ct_sel(k_ok, ST_PENDING, ERR_BAD_RNG_V), ERR_BAD_PRIVKEY_V);
}

// x mod q for x of NX limbs (NX <= 3N/2 + 1), result N limbs in [0, q).  2^(32N) = cq (mod q), cq of N/2 + 1 limbs.
template <int N, int NX>
__device__ __forceinline__ void mod_q_ct(uint32_t (&r)[N], const uint32_t (&x)[NX])
{
    constexpr int NC = N / 2 + 1, NH = NX - N;         // limbs of cq, of the high part
    static_assert(NH >= 1 && NH <= N / 2 + 1, "operand size");
    uint32_t cq[NC], q[N];
#pragma unroll
    for (int i = 0; i < NC; ++i) cq[i] = curve_cq<N>()[i];
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = curve_q<N>()[i];
    // v = lo + hi * cq : up to N + 1 limbs (hi * cq < 2^(32N + 2))
    uint32_t v[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; ++i) v[i] = i < N ? x[i] : 0u;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (i + j < N + 2) {
                c += (uint64_t)x[N + i] * cq[j] + v[i + j];
                v[i + j] = (uint32_t)c; c >>= 32;
            }
        }
#pragma unroll
        for (int j = i + NC; j < N + 2; ++j) { c += v[j]; v[j] = (uint32_t)c; c >>= 32; }
    }
    // fold the top (v[N], v[N+1] small) twice more: t * cq + lo
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t t = v[N];                       // < 8 after the first pass, 0 or 1 after the second
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N + 1; ++j) {
            c += (uint64_t)(j < N ? v[j] : 0u) + (j < NC ? (uint64_t)t * cq[j] : 0u);
            v[j] = (uint32_t)c; c >>= 32;
        }
    }
    // v < 2^(32N) = q + cq now (v[N] == 0): one masked subtraction of q
    uint32_t s[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t dd = (uint64_t)v[i] - q[i] - borrow;
        s[i] = (uint32_t)dd; borrow = (uint32_t)(dd >> 32) & 1u;
    }
    const uint32_t lt = 0u - borrow;                   // v < q: keep v
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = ct_sel(lt, v[i], s[i]);
}
// zzSubMod (zz_mod.c:120-132): c = a - b, + q when the subtraction borrowed.  b need not be below q.
template <int N>
__device__ __forceinline__ void sub_mod_q_ct(uint32_t (&c)[N], const uint32_t (&a)[N], const uint32_t (&b)[N])
{
    uint32_t t[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
    }
    const uint32_t m = 0u - borrow;
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cy += (uint64_t)t[i] + (curve_q<N>()[i] & m);
        c[i] = (uint32_t)cy; cy >>= 32;
    }
}

// s0 = belt-hash(oid || <x_R> || H)[0 .. l bits), s1 = (k - (s0 + 2^l) d - H) mod q (bign_sign.c:221-238);
// wipes k.  x_R and H are public, so this hash could use any table; it shares BeltTabTwo with the nonce kernel.
template <int N>
__global__ __launch_bounds__(1024)
void bign_sign_tail_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ privkeys,
                           const uint8_t *__restrict__ rx, uint8_t *__restrict__ ks, size_t n, OidArg oid,
                           const uint32_t *__restrict__ status, uint8_t *__restrict__ sigs, uint32_t *__restrict__ codes,
                           uint32_t row_words)
{
    constexpr int NO = 4 * N;
    const uint32_t ROW = row_words;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint8_t *s_tab = s_dyn;
    uint32_t *s_rows = reinterpret_cast<uint32_t *>(s_dyn + BeltTabTwo::kBytes);
    BeltTabTwo::fill(s_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const BeltTabTwoP T(s_tab);         // (round 4) the SDWA-address form of the same bank-private table: 8 instead of 12 VALU per G-box
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t st = status[idx];
    const uint32_t ok = ct_eq_small(st >> 1, ST_PENDING >> 1);      // status is public (an error code the caller sees)

    uint32_t xr[N], H[N], d[N], k[N];
    load_words_bytes(xr, rx + NO * idx);
    load_words_bytes(H, hashes + NO * idx);
    load_words_bytes(d, privkeys + NO * idx);
    load_words_bytes(k, ks + NO * idx);
    uint32_t *kz = reinterpret_cast<uint32_t *>(ks + NO * idx);
#pragma unroll
    for (int i = 0; i < N; ++i) kz[i] = 0;                           // the one-time key does not outlive the call

    uint32_t *row = s_rows + threadIdx.x * ROW;
    const uint32_t len = oid.len + 2 * NO;
    const uint32_t nwords = (len + 31) / 32 * 8;
    for (uint32_t i = 0; i < nwords; ++i) row[i] = 0;
    for (uint32_t i = 0; i < oid.len; ++i) row_put_bytes(row, i, oid.der[i]);
    {
        const uint32_t sh = (oid.len & 3u) * 8u, w0 = oid.len >> 2;
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) {
            const uint32_t v = i < N ? xr[i] : H[i - N];
            row[w0 + i] |= v << sh;
            if (sh) row[w0 + i + 1] |= v >> (32 - sh);
        }
    }
    uint32_t h[8];
    belt_hash_row(T, h, row, len, oid);

    // (s0 + 2^l) d : (N/2 + 1) x N limbs
    uint32_t s0[N / 2 + 1];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) s0[i] = h[i];
    s0[N / 2] = 1u;
    uint32_t prod[N + N / 2 + 1];
#pragma unroll
    for (int i = 0; i < N + N / 2 + 1; ++i) prod[i] = 0;
#pragma unroll
    for (int i = 0; i <= N / 2; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            c += (uint64_t)s0[i] * d[j] + prod[i + j];
            prod[i + j] = (uint32_t)c; c >>= 32;
        }
        prod[i + N] += (uint32_t)c;
    }
    uint32_t s1[N];
    mod_q_ct<N, N + N / 2 + 1>(s1, prod);
    sub_mod_q_ct<N>(s1, k, s1);
    sub_mod_q_ct<N>(s1, s1, H);

    uint32_t *so = reinterpret_cast<uint32_t *>(sigs + (NO + NO / 2) * idx);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) so[i] = h[i] & ok;
#pragma unroll
    for (int i = 0; i < N; ++i) so[N / 2 + i] = s1[i] & ok;
    codes[idx] = ct_sel(ok, (uint32_t)ERR_OK, st);
}

// ------------------------------------------------------------------ host side ---
struct SignScratch { uint32_t *status; uint8_t *k; uint8_t *rx; };
template <int N>
static err_t sign_scratch(hipStream_t st, size_t n, SignScratch &S)
{
    const size_t n_pad = (n + 63) & ~(size_t)63;
    void *base = nullptr;
    err_t code = scratch_for_stream(st, 8 + N / 4 - 2, n_pad * (4 + 4 * N + 4 * N), &base);
    if (code != ERR_OK) return code;
    S.status = (uint32_t *)base;
    S.k = (uint8_t *)base + 4 * n_pad;
    S.rx = S.k + 4 * N * n_pad;
    return ERR_OK;
}

// k G by one lane per scalar (throughput) or by 64 / 16 / 4 lanes per scalar (latency; each form fills the device -- one
// wavefront per SIMD -- at 2^10 / 2^12 / 2^14 scalars).  g_sign_lanes: 0 = by batch size; 1 / 102 / 101 = always one lane
// (signed 6-bit windows with Jacobian mixed additions / with complete additions / unsigned 4-bit windows), 4 / 16 / 64 forced
// (bee2hip_internal_tune 10).  tools/ab/sign_coop_ab.py measures all six at every size on the three curves
// (profiles/r03_sign_coop.txt): 64 lanes up to 2^10 scalars, 16 up to 2^13, 4 up to 2^15, one lane above.
static int g_sign_lanes = 0;
void set_sign_coop(int v) { g_sign_lanes = v; }
// (round 4) 8 / 7 = one lane per scalar with the window's entry looked up in LDS (bign_mulbase_lds_kernel: 256-bit curve only,
// workgroups of 1024 lanes; signed 8-bit windows and 16 copies of the row -- the product -- or signed 7-bit windows and 32
// copies): from 3 * 2^16 scalars on, where its one round of <= 256 workgroups (0.63 ms) beats the scanning kernel's 256-lane
// blocks (0.55 ms at 2^17, 0.97 ms at 2^18: tools/ab/sign_lds_ab.py)
constexpr size_t MULBASE_LDS_MIN = (size_t)1 << 15, MULBASE_LDS_512_MAX = (size_t)1 << 17;   // (512-lane workgroups; round 6: from 2^15 scalars -- the 29-bit form with the shared inversion
//  takes 0.27 | 0.35 ms (key | signature) for anything up to 2^17, the 4-lane cooperative form 0.35 | 0.43 ms at 2^15: profiles/r06_sign_l29_ab.txt)
constexpr bool LDS_XYZZ = false;       // accumulator of the 7-bit LDS form: Jacobian (8M + 3S).  XYZZ (8M + 2S, one more coordinate) measured: +-0 % (profiles/r04_sign_lds.txt)
template <int N>
static inline int mulbase_lanes(size_t n)
{
    if (g_sign_lanes == 1 || g_sign_lanes == 4 || g_sign_lanes == 16 || g_sign_lanes == 64 || g_sign_lanes == 101 || g_sign_lanes == 102)
        return g_sign_lanes;
    if (N == 8 && (g_sign_lanes == 72 || g_sign_lanes == 7)) return g_sign_lanes;
    if (N == 8 && (g_sign_lanes == 8 || (g_sign_lanes == 0 && n >= MULBASE_LDS_MIN))) return 8;
#ifdef BEE2HIP_EXPERIMENTS
    if (N == 8 && g_sign_lanes == 81) return 81;            // the LDS look-up form on 32-bit limbs (round 4-5): A/B record against the 29-bit one
#endif
    return n <= ((size_t)1 << 10) ? 64 : n <= ((size_t)1 << 13) ? 16 : n <= ((size_t)1 << 15) ? 4 : 1;
}
// the table the chosen form reads: the signed 7-bit one for form 7, the signed 6-bit one otherwise (`tab6` of launch_mulbase)
template <int N>
static err_t mulbase_tables(int lanes, const uint32_t **tab, const uint32_t **tabw, hipStream_t st)
{
    err_t code = bign_table6<N>(tab, tabw, st);
    if (code == ERR_OK && (lanes == 7 || lanes == 72 || lanes == 8 || lanes == 81)) {
        if constexpr (N != 8) return ERR_BAD_INPUT;                      // (never: mulbase_lanes picks these forms on the 256-bit curve only)
        else {
#ifdef BEE2HIP_EXPERIMENTS
        code = lanes == 8 || lanes == 81 ? bign_tablew<N, 8>(tabw, st) : bign_tablew<N, 7>(tabw, st);
#else
        code = bign_tablew<N, 8>(tabw, st);
#endif
        }
        if (code == ERR_OK) {
            static std::once_flag once[64];
            hipError_t e = hipSuccess;
            std::call_once(once[cur_dev() & 63], [&] {
                const int bytes = (int)((size_t)WinW<8, 7>::ENT * 8 * 32 * 8);          // = 128 entries x 64 octets x 16 copies too
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(bign_mulbase_lds_kernel<8, 8, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void *>(bign_mulbase_lds_kernel<8, 8, false, 512>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#ifdef BEE2HIP_EXPERIMENTS
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void *>(bign_mulbase_lds_kernel<8, 8, false, 1024, false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void *>(bign_mulbase_lds_kernel<8, 8, false, 512, false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void *>(bign_mulbase_lds_kernel<8, 7, LDS_XYZZ>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void *>(bign_mulbase_lds_kernel<8, 7, !LDS_XYZZ>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#endif
            });
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(bign_mulbase_lds_kernel)");
        }
    }
    return code;
}
// lanes: 64 / 16 / 4 = cooperative forms on the 4-bit windows of the seed table; 1 = one lane per scalar, signed 6-bit windows
// (tab6), Jacobian mixed additions; 102 = the same windows with complete additions; 101 = one lane per scalar on the 4-bit
// windows (the round-2 kernel) -- the last two kept for A/B and as second opinions in the tests
template <int N, int MODE, bool X_ONLY>
static void launch_mulbase(int lanes, const uint8_t *scalars, size_t n, uint32_t *codes, uint8_t *out, const uint32_t *tab,
                           const uint32_t *tab6, hipStream_t st)
{
    const dim3 g256((unsigned)((n + 255) / 256));
    if (lanes == 64 || lanes == 16 || lanes == 4)
        hipLaunchKernelGGL((bign_mulbase_coop_kernel<N>), dim3((unsigned)((n * (size_t)lanes + 63) / 64)), dim3(64), 0, st, scalars, n,
                           codes, out, tab, MODE, (int)X_ONLY, lanes);
#ifdef BEE2HIP_EXPERIMENTS      // second opinions of the tests and the A/B record (tools/ab/sign_coop_ab.py)
    else if (lanes == 101)
        hipLaunchKernelGGL((bign_mulbase_ct_kernel<N, 0>), g256, dim3(256), 0, st, scalars, n, codes, out, tab, MODE, (int)X_ONLY);
    else if (lanes == 102)
        hipLaunchKernelGGL((bign_mulbase_ct_kernel<N, 1>), g256, dim3(256), 0, st, scalars, n, codes, out, tab6, MODE, (int)X_ONLY);
#endif
#ifdef BEE2HIP_EXPERIMENTS
    else if (lanes == 81) {
        if constexpr (N == 8) {
            constexpr size_t lds = (size_t)WinW<N, 8>::ENT * N * 8 * 16;
            if (n <= MULBASE_LDS_512_MAX)
                hipLaunchKernelGGL((bign_mulbase_lds_kernel<N, 8, false, 512, false>), dim3((unsigned)((n + 511) / 512)), dim3(512), lds, st, scalars,
                                   n, codes, out, reinterpret_cast<const uint64_t *>(tab6), MODE, (int)X_ONLY);
            else
                hipLaunchKernelGGL((bign_mulbase_lds_kernel<N, 8, false, 1024, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(1024), lds, st, scalars, n,
                                   codes, out, reinterpret_cast<const uint64_t *>(tab6), MODE, (int)X_ONLY);
        }
    }
#endif
    else if (lanes == 8) {
        // signed 8-bit windows, 16 copies of the row read with ds_read_b128 (tab6 = the 8-bit window table): 33 additions
        if constexpr (N == 8) {
            constexpr size_t lds = (size_t)WinW<N, 8>::ENT * N * 8 * 16;
            // up to 2^17 scalars: workgroups of 512 lanes (<= 256 of them: one per CU, two wavefronts per SIMD); above: 1024 lanes
            if (n <= MULBASE_LDS_512_MAX)
                hipLaunchKernelGGL((bign_mulbase_lds_kernel<N, 8, false, 512>), dim3((unsigned)((n + 511) / 512)), dim3(512), lds, st, scalars,
                                   n, codes, out, reinterpret_cast<const uint64_t *>(tab6), MODE, (int)X_ONLY);
            else
                hipLaunchKernelGGL((bign_mulbase_lds_kernel<N, 8, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(1024), lds, st, scalars, n,
                                   codes, out, reinterpret_cast<const uint64_t *>(tab6), MODE, (int)X_ONLY);
        }
    }
    else if (lanes == 7 || lanes == 72) {
        // 256-bit curve, batches that give every CU a workgroup of 1024 lanes: signed 7-bit windows looked up in LDS (tab6 = that table)
        if constexpr (N == 8) {
            constexpr size_t lds = (size_t)WinW<N, 7>::ENT * N * 32 * 8;
            const dim3 g((unsigned)((n + 1023) / 1024));
            (void)lds; (void)g;
#ifdef BEE2HIP_EXPERIMENTS      // the 7-bit forms: A/B record and second opinions in the tests
            if (lanes == 7)
                hipLaunchKernelGGL((bign_mulbase_lds_kernel<N, 7, LDS_XYZZ>), g, dim3(1024), lds, st, scalars, n, codes, out,
                                   reinterpret_cast<const uint64_t *>(tab6), MODE, (int)X_ONLY);
            else
                hipLaunchKernelGGL((bign_mulbase_lds_kernel<N, 7, !LDS_XYZZ>), g, dim3(1024), lds, st, scalars, n, codes, out,
                                   reinterpret_cast<const uint64_t *>(tab6), MODE, (int)X_ONLY);
#endif
        }
    }
    else
        hipLaunchKernelGGL((bign_mulbase_ct_kernel<N, 2>), g256, dim3(256), 0, st, scalars, n, codes, out, tab6, MODE, (int)X_ONLY);
}

template <int N>
static err_t launch_bign_pubkey_calc_t(bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes, hipStream_t st)
{
    const uint32_t *tab = nullptr, *tab6 = nullptr;
    const int lanes = mulbase_lanes<N>(n);
    err_t code = mulbase_tables<N>(lanes, &tab, &tab6, st);
    if (code != ERR_OK) return code;
    if (!keygen)
        launch_mulbase<N, 1, false>(lanes, (const uint8_t *)d_privkeys, n, (uint32_t *)d_codes, (uint8_t *)d_pubkeys, tab, tab6, st);
    else
        launch_mulbase<N, 2, false>(lanes, (const uint8_t *)d_privkeys, n, (uint32_t *)d_codes, (uint8_t *)d_pubkeys, tab, tab6, st);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// LDS row of the hashing kernels: the message zero-padded to whole 32-octet blocks, as an odd number of 32-bit words
static inline uint32_t sign_row_words(size_t msg_octets) { return (uint32_t)((msg_octets + 31) / 32 * 8) | 1u; }
// lanes per workgroup (one 64 KiB table each): big batches take the largest that fits the CU's 160 KiB with its rows
// (256-bit curve, no t: 1024 lanes for the nonce, 512 for the tail -- 4 and 2 wavefronts per SIMD where it used to be 1)
static int g_sign_wg = 0;                                  // bee2hip_internal_tune 12: 0 = by size, 256 / 512 / 1024 = upper bound
void set_sign_wg(int v) { g_sign_wg = v; }
static inline int sign_wg(size_t n, uint32_t row_words)
{
    // ... as long as every CU still gets a workgroup: 2^16 signatures in 64 workgroups of 1024 lose 8 % (tools/ab/sign_wg_ab.py)
    if (n < ((size_t)1 << 17)) return SIGN_WG;
    const int fit = n < ((size_t)1 << 18) ? 512 : 1024;
    const int cap = g_sign_wg == 256 || g_sign_wg == 512 || g_sign_wg == 1024 ? (g_sign_wg < fit ? g_sign_wg : fit) : fit;
    for (int wg = cap; wg > SIGN_WG; wg >>= 1)
        if (BeltTabTwo::kBytes + (size_t)wg * row_words * 4 <= (size_t)160 * 1024) return wg;
    return SIGN_WG;
}

// mode 0: deterministic (bignSign2): d_aux = t (n x t_len, or shared when t_shared), may be null
// mode 1: one-time keys supplied (bignSign after its rng): d_aux = k, n x 4N octets
// mode 2: deterministic with theta = belt-hash(oid || d || t) supplied: d_aux = n x 32 octets
template <int N>
static err_t launch_bign_sign_t(int mode, const uint8_t *oid_der, size_t oid_len, const void *d_hashes, const void *d_privkeys,
                                const void *d_aux, size_t t_len, int t_shared, size_t n, void *d_sigs, void *d_codes,
                                hipStream_t st)
{
    // mode 2: d_aux = theta (n x 32 octets), computed by the caller
    if (n == 0) return ERR_OK;
    if (mode == 0 && d_aux && t_len > (size_t)SIGN_T_MAX) return ERR_NOT_IMPLEMENTED;
    OidArg oa;
    err_t code = make_oid_arg(oa, oid_der, oid_len, st);
    if (code != ERR_OK) return code;
    const uint32_t *tab = nullptr, *tab6 = nullptr;
    const int lanes = mulbase_lanes<N>(n);
    code = mulbase_tables<N>(lanes, &tab, &tab6, st);
    if (code != ERR_OK) return code;
    SignScratch S;
    code = sign_scratch<N>(st, n, S);
    if (code != ERR_OK) return code;
    const uint8_t *kptr;
    QArg<N> qa;
    memset(&qa, 0, sizeof qa);
    qa.std = 1;
    if (mode == 0 || mode == 2) {
        const uint8_t *tp = mode == 0 ? (const uint8_t *)d_aux : nullptr;
        const uint32_t tl = (uint32_t)(tp ? t_len : 0);
        const uint32_t rw = sign_row_words(oa.len + 4 * N + tl);
        const int wg = sign_wg(n, rw);
        const size_t lds = BeltTabTwo::kBytes + (size_t)wg * rw * 4;
        const auto go = [&](auto kern) -> err_t {
            B2H_TRY(dyn_lds_once((const void *)kern, lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)((n + wg - 1) / wg)), dim3(wg), lds, st, (const uint8_t *)d_hashes,
                               (const uint8_t *)d_privkeys, tp, tl, (uint32_t)(t_shared ? 0 : t_len),
                               mode == 2 ? (const uint8_t *)d_aux : nullptr, n, oa, qa, S.status, S.k, rw);
            return ERR_OK;
        };
        code = go(bign_sign_nonce_kernel<N>);
        if (code != ERR_OK) return code;
        kptr = S.k;
    } else {
        hipLaunchKernelGGL(bign_sign_kcheck_kernel<N>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_privkeys, (const uint8_t *)d_aux, n, qa, S.status);
        B2H_TRY(hipMemcpyAsync(S.k, d_aux, n * 4 * N, hipMemcpyDeviceToDevice, st));
        kptr = S.k;
    }
    launch_mulbase<N, 0, true>(lanes, kptr, n, (uint32_t *)nullptr, S.rx, tab, tab6, st);
    {
        const uint32_t rw = sign_row_words(oa.len + 8 * N);
        const int wg = sign_wg(n, rw);
        const size_t lds = BeltTabTwo::kBytes + (size_t)wg * rw * 4;
        const auto go = [&](auto kern) -> err_t {
            B2H_TRY(dyn_lds_once((const void *)kern, lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)((n + wg - 1) / wg)), dim3(wg), lds, st, (const uint8_t *)d_hashes,
                               (const uint8_t *)d_privkeys, (const uint8_t *)S.rx, S.k, n, oa, (const uint32_t *)S.status,
                               (uint8_t *)d_sigs, (uint32_t *)d_codes, rw);
            return ERR_OK;
        };
        code = go(bign_sign_tail_kernel<N>);
        if (code != ERR_OK) return code;
    }
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

err_t launch_bign_pubkey_calc(size_t l, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    switch (l) {
    case 128: return launch_bign_pubkey_calc_t<8>(keygen, d_privkeys, n, d_pubkeys, d_codes, st);
    case 192: return launch_bign_pubkey_calc_t<12>(keygen, d_privkeys, n, d_pubkeys, d_codes, st);
    case 256: return launch_bign_pubkey_calc_t<16>(keygen, d_privkeys, n, d_pubkeys, d_codes, st);
    default: return ERR_BAD_PARAMS;
    }
}
err_t launch_bign_sign(size_t l, int mode, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                       const void *d_privkeys, const void *d_aux, size_t t_len, int t_shared, size_t n, void *d_sigs,
                       void *d_codes, hipStream_t st)
{
    switch (l) {
    case 128: return launch_bign_sign_t<8>(mode, oid_der, oid_len, d_hashes, d_privkeys, d_aux, t_len, t_shared, n, d_sigs, d_codes, st);
    case 192: return launch_bign_sign_t<12>(mode, oid_der, oid_len, d_hashes, d_privkeys, d_aux, t_len, t_shared, n, d_sigs, d_codes, st);
    case 256: return launch_bign_sign_t<16>(mode, oid_der, oid_len, d_hashes, d_privkeys, d_aux, t_len, t_shared, n, d_sigs, d_codes, st);
    default: return ERR_BAD_PARAMS;
    }
}

}  // namespace bee2hip
