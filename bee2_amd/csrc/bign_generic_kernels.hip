// bign_generic_kernels.hip -- bignVerify / bignPubkeyVal for NON-STANDARD parameter sets (VERDICT r01 "missing" 4).
//
// The reference accepts whatever passes bignParamsCheck + bignEcCreate (src/crypto/bign/bign_params.c:244-280,
// bign_ec.c:29-80): any odd 2l-bit p = 3 (mod 4), coefficients a, b < p, base point (0, yG), odd 2l-bit q.  The three
// standard curves have their own kernels (bign_kernels.hip: p = 2^(32N) - c, a = -3, comb tables of G); everything
// else comes here.  This is a completeness path, not a throughput path -- one lane per signature, no tables:
//   * GF(p) in Montgomery form on N = l/16 32-bit limbs (CIOS, what zmMulMont / zzRedMont do on 64-bit words,
//     src/math/zm.c:129-212), values kept canonical (< p); inversion a^(p-2) as gfpInv (gfp.c:33-44);
//   * Jacobian points with a general coefficient a (ecpDblJ, ecp_j.c:241-299: 3 X^2 + a Z^4) and the complete
//     addition with every exceptional case handled inline (ecpAddJ, ecp_j.c:397-497);
//   * R = u G + v Q by one simultaneous double-and-add over the 2l bits of u and the l + 1 bits of v
//     (ecAddMulA's result for points on the curve, ec.c:1183-1273);
//   * range checks, u = (s1 + H) mod q, v = s0 + 2^l exactly as bignVerifyEc (bign_sign.c:306-330);
//   * x_R goes to the scratch of the standard pipeline and bign_tail_kernel finishes (belt-hash, comparison).
// As for the standard curves, parity with the reference is structural for a prime p and keys on the curve; for
// off-curve keys (never validated by bignVerify) both sides reject with probability 1 - 2^-l.
//
// Round 3 (VERDICT r02 "missing" 4): the SIGNING side on non-standard sets -- bignPubkeyCalc, bignKeypairGen, bignSign,
// bignSign2 (src/crypto/bign/bign_misc.c:182-243,373-431, bign_sign.c:32-260).  The scalars are secret, so this part is
// constant-time at the instruction level like bign_sign_kernels.hip:
//   * k G by double-and-add-ALWAYS over all 2l bits with the COMPLETE addition of Renes, Costello and Batina for a
//     general coefficient a (algorithm 1: homogeneous projective, 12 M + 3 m_a + 2 m_3b, no exceptional case: O, P = Q
//     and P = -Q go through the same instructions; modelled against affine arithmetic in tools/model_rcb_general.py),
//     the doubling being the same addition with both operands equal; the bit selects the sum with masks;
//   * the field routines above contain selections (v_cndmask), no branches; the only loops with data-dependent trip counts
//     are over PUBLIC data (the exponent p - 2 of the inversion);
//   * arithmetic mod q in Montgomery form (the same g_mul with q as the modulus): s1 = k - H - (s0 + 2^l) d;
//   * nonce derivation, range checks and the hash of oid || x_R || H are the kernels / device functions of
//     bign_sign_kernels.hip with q passed as an argument.
// One lane per item, no tables: about 2 x 2l complete additions per scalar multiplication -- a completeness path.
#include "common.hpp"

namespace bee2hip {

template <int N> struct GenCurve {
    uint32_t p[N], q[N];
    uint32_t a[N], b[N], gy[N];     // Montgomery form
    uint32_t one[N];                // R mod p
    uint32_t r2[N];                 // R^2 mod p
    uint32_t pm2[N];                // p - 2
    uint32_t n0;                    // -p^-1 mod 2^32
};

template <int N> struct gfe { uint32_t v[N]; };

template <int N>
__device__ __forceinline__ bool g_ge(const uint32_t (&a)[N], const uint32_t (&b)[N])
{
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        borrow = (uint32_t)(d >> 32) & 1u;
    }
    return borrow == 0;
}
template <int N>
__device__ __forceinline__ bool g_is_zero(const gfe<N> &a)
{
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z |= a.v[i];
    return z == 0;
}
template <int N>
__device__ __forceinline__ bool g_eq(const gfe<N> &a, const gfe<N> &b)
{
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z |= a.v[i] ^ b.v[i];
    return z == 0;
}
// r = a + b mod p (a, b < p)
template <int N>
__device__ __noinline__ void g_add(gfe<N> &r, const gfe<N> &a, const gfe<N> &b, const GenCurve<N> &C)
{
    uint32_t t[N], s[N];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { c += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)t[i] - C.p[i] - borrow;
        s[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
    }
    const bool ge = c != 0 || borrow == 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = ge ? s[i] : t[i];
}
template <int N>
__device__ __noinline__ void g_sub(gfe<N> &r, const gfe<N> &a, const gfe<N> &b, const GenCurve<N> &C)
{
    uint32_t t[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
        t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
    }
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (uint64_t)t[i] + (borrow ? C.p[i] : 0u);
        r.v[i] = (uint32_t)c; c >>= 32;
    }
}
// Montgomery product a b R^-1 mod p (CIOS), a, b < p -> result < p
template <int N>
__device__ __noinline__ void g_mul(gfe<N> &r, const gfe<N> &a, const gfe<N> &b, const GenCurve<N> &C)
{
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint64_t s = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)s; c = s >> 32;
        }
        uint64_t s = (uint64_t)t[N] + c;
        t[N] = (uint32_t)s; t[N + 1] = (uint32_t)(s >> 32);
        const uint32_t m = t[0] * C.n0;
        s = (uint64_t)m * C.p[0] + t[0];
        c = s >> 32;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            s = (uint64_t)m * C.p[j] + t[j] + c;
            t[j - 1] = (uint32_t)s; c = s >> 32;
        }
        s = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)s;
        t[N] = t[N + 1] + (uint32_t)(s >> 32);
    }
    uint32_t s2[N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)t[i] - C.p[i] - borrow;
        s2[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
    }
    const bool ge = t[N] != 0 || borrow == 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = ge ? s2[i] : t[i];
}
template <int N>
__device__ __forceinline__ void g_sqr(gfe<N> &r, const gfe<N> &a, const GenCurve<N> &C) { g_mul(r, a, a, C); }
template <int N>
__device__ __forceinline__ void g_set(gfe<N> &r, const uint32_t (&w)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = w[i];
}
// a^(p-2) (Montgomery domain in, Montgomery domain out); 0 -> 0
template <int N>
__device__ __noinline__ void g_inv(gfe<N> &r, const gfe<N> &a, const GenCurve<N> &C)
{
    gfe<N> acc;
    g_set(acc, C.one);
#pragma unroll 1
    for (int i = 32 * N - 1; i >= 0; --i) {
        g_sqr(acc, acc, C);
        if ((C.pm2[i >> 5] >> (i & 31)) & 1u) g_mul(acc, acc, a, C);
    }
    r = acc;
}

template <int N> struct gjac { gfe<N> X, Y, Z; };          // O <=> Z == 0

// T <- 2T for y^2 = x^3 + a x + b (ecpDblJ): M = 3 X^2 + a Z^4, S = 4 X Y^2
template <int N>
__device__ __noinline__ void gj_dbl(gjac<N> &T, const GenCurve<N> &C)
{
    if (g_is_zero(T.Z)) return;
    if (g_is_zero(T.Y)) { for (int i = 0; i < N; ++i) T.Z.v[i] = 0; return; }
    gfe<N> XX, YY, ZZ, S, M, t, a;
    g_set(a, C.a);
    g_sqr(XX, T.X, C);
    g_sqr(YY, T.Y, C);
    g_sqr(ZZ, T.Z, C);
    g_mul(S, T.X, YY, C);
    g_add(S, S, S, C); g_add(S, S, S, C);             // 4 X Y^2
    g_sqr(ZZ, ZZ, C);
    g_mul(ZZ, ZZ, a, C);                              // a Z^4
    g_add(M, XX, XX, C); g_add(M, M, XX, C);
    g_add(M, M, ZZ, C);
    g_mul(T.Z, T.Y, T.Z, C);
    g_add(T.Z, T.Z, T.Z, C);                          // Z3 = 2 Y Z
    g_sqr(t, M, C);
    g_sub(t, t, S, C);
    g_sub(T.X, t, S, C);                              // X3 = M^2 - 2 S
    g_sqr(YY, YY, C);
    g_add(YY, YY, YY, C); g_add(YY, YY, YY, C); g_add(YY, YY, YY, C);   // 8 Y^4
    g_sub(t, S, T.X, C);
    g_mul(t, M, t, C);
    g_sub(T.Y, t, YY, C);                             // Y3 = M (S - X3) - 8 Y^4
}
// T <- T + E, every case (ecpAddJ)
template <int N>
__device__ __noinline__ void gj_add(gjac<N> &T, const gjac<N> &E, const GenCurve<N> &C)
{
    if (g_is_zero(E.Z)) return;
    if (g_is_zero(T.Z)) { T = E; return; }
    gfe<N> Z1Z1, Z2Z2, U1, U2, S1, S2, H, r, HH, HHH, V, t;
    g_sqr(Z1Z1, T.Z, C);
    g_sqr(Z2Z2, E.Z, C);
    g_mul(U1, T.X, Z2Z2, C);
    g_mul(U2, E.X, Z1Z1, C);
    g_mul(t, E.Z, Z2Z2, C); g_mul(S1, T.Y, t, C);
    g_mul(t, T.Z, Z1Z1, C); g_mul(S2, E.Y, t, C);
    g_sub(H, U2, U1, C);
    g_sub(r, S2, S1, C);
    if (g_is_zero(H)) {
        if (g_is_zero(r)) gj_dbl(T, C);               // T == E
        else for (int i = 0; i < N; ++i) T.Z.v[i] = 0;   // T == -E
        return;
    }
    g_sqr(HH, H, C);
    g_mul(HHH, H, HH, C);
    g_mul(V, U1, HH, C);
    g_mul(t, T.Z, E.Z, C); g_mul(T.Z, t, H, C);
    g_sqr(t, r, C);
    g_sub(t, t, HHH, C);
    g_sub(t, t, V, C);
    g_sub(T.X, t, V, C);
    g_sub(t, V, T.X, C);
    g_mul(t, r, t, C);
    g_mul(S2, S1, HHH, C);
    g_sub(T.Y, t, S2, C);
}

// one signature: the whole of bignVerifyEc up to x_R.  Status: a final err_t, or ST_PENDING with canonical x_R in rx.
template <int N>
__global__ __launch_bounds__(64)
void bign_generic_verify_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                                const uint8_t *__restrict__ pubkeys, size_t n, VerifyScratch S, GenCurve<N> C)
{
    constexpr int NO = 4 * N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    gjac<N> Q, G, T;
    uint32_t s1[N], H[N], u[N], v[N / 2 + 1];
    const uint8_t *sig = sigs + (NO + NO / 2) * idx;
    for (int i = 0; i < N; ++i) {
        Q.X.v[i] = reinterpret_cast<const uint32_t *>(pubkeys + 2 * NO * idx)[i];
        Q.Y.v[i] = reinterpret_cast<const uint32_t *>(pubkeys + 2 * NO * idx + NO)[i];
        s1[i] = reinterpret_cast<const uint32_t *>(sig + NO / 2)[i];
        H[i] = reinterpret_cast<const uint32_t *>(hashes + NO * idx)[i];
    }
    // qrFrom: coordinates < p (bign_sign.c:306-311); s1 < q (:313-318)
    if (g_ge(Q.X.v, C.p) || g_ge(Q.Y.v, C.p)) { S.status[idx] = ERR_BAD_PUBKEY; return; }
    if (g_ge(s1, C.q)) { S.status[idx] = ERR_BAD_SIG; return; }
    // H <- H - q if H >= q; u <- (s1 + H) mod q (:320-327)
    {
        uint32_t t[N];
        uint32_t borrow = 0;
        for (int i = 0; i < N; ++i) {
            const uint64_t d = (uint64_t)H[i] - C.q[i] - borrow;
            t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
        }
        for (int i = 0; i < N; ++i) H[i] = borrow ? H[i] : t[i];
        uint64_t c = 0;
        uint32_t s[N];
        for (int i = 0; i < N; ++i) { c += (uint64_t)s1[i] + H[i]; s[i] = (uint32_t)c; c >>= 32; }
        const uint32_t carry = (uint32_t)c;
        borrow = 0;
        for (int i = 0; i < N; ++i) {
            const uint64_t d = (uint64_t)s[i] - C.q[i] - borrow;
            t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
        }
        const bool ge = carry || !borrow;
        for (int i = 0; i < N; ++i) u[i] = ge ? t[i] : s[i];
    }
    for (int i = 0; i < N / 2; ++i) v[i] = reinterpret_cast<const uint32_t *>(sig)[i];
    v[N / 2] = 1u;                                   // + 2^l (:329-330)

    gfe<N> r2;
    g_set(r2, C.r2);
    g_mul(Q.X, Q.X, r2, C);
    g_mul(Q.Y, Q.Y, r2, C);
    g_set(Q.Z, C.one);
    for (int i = 0; i < N; ++i) G.X.v[i] = 0;
    g_set(G.Y, C.gy);
    g_set(G.Z, C.one);
    for (int i = 0; i < N; ++i) { T.X.v[i] = 0; T.Z.v[i] = 0; }
    g_set(T.Y, C.one);
#pragma unroll 1
    for (int i = 32 * N - 1; i >= 0; --i) {
        gj_dbl(T, C);
        if ((u[i >> 5] >> (i & 31)) & 1u) gj_add(T, G, C);
        if (i <= 16 * N && ((v[i >> 5] >> (i & 31)) & 1u)) gj_add(T, Q, C);
    }
    if (g_is_zero(T.Z)) { S.status[idx] = ERR_BAD_SIG; return; }     // R == O (:332-336)
    gfe<N> zi, x, one;
    g_inv(zi, T.Z, C);
    g_sqr(zi, zi, C);
    g_mul(x, T.X, zi, C);
    for (int i = 0; i < N; ++i) one.v[i] = i == 0;
    g_mul(x, x, one, C);                             // out of the Montgomery domain: canonical x_R
    for (int l = 0; l < N; ++l) S.rx[(size_t)l * S.n_pad + idx] = x.v[l];
    S.status[idx] = ST_PENDING;
}

// bignPubkeyValEc for a general curve: coordinates < p and y^2 == x^3 + a x + b (ecpIsOnA, ecp_a.c:36-60)
template <int N>
__global__ __launch_bounds__(64)
void bign_generic_pubkey_val_kernel(const uint8_t *__restrict__ pubkeys, size_t n, uint32_t *__restrict__ codes, GenCurve<N> C)
{
    constexpr int NO = 4 * N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    gfe<N> x, y, t, a, b, r2;
    for (int i = 0; i < N; ++i) {
        x.v[i] = reinterpret_cast<const uint32_t *>(pubkeys + 2 * NO * idx)[i];
        y.v[i] = reinterpret_cast<const uint32_t *>(pubkeys + 2 * NO * idx + NO)[i];
    }
    if (g_ge(x.v, C.p) || g_ge(y.v, C.p)) { codes[idx] = ERR_BAD_PUBKEY; return; }
    g_set(r2, C.r2); g_set(a, C.a); g_set(b, C.b);
    g_mul(x, x, r2, C);
    g_mul(y, y, r2, C);
    g_sqr(t, x, C);
    g_add(t, t, a, C);
    g_mul(t, t, x, C);
    g_add(t, t, b, C);
    g_sqr(y, y, C);
    codes[idx] = g_eq(t, y) ? ERR_OK : ERR_BAD_PUBKEY;
}


// ------------------------------------------------------------ signing side ---
template <int N> struct gproj { gfe<N> X, Y, Z; };          // x = X / Z; O = (0 : 1 : 0)

// R <- P + Q, complete (Renes-Costello-Batina algorithm 1, general a; b3 = 3 b).  R may be P or Q.
template <int N>
__device__ __noinline__ void gp_add_complete(gproj<N> &R, const gproj<N> &P, const gproj<N> &Q, const gfe<N> &a, const gfe<N> &b3,
                                             const GenCurve<N> &C)
{
    gfe<N> t0, t1, t2, t3, t4, t5, X3, Y3, Z3;
    g_mul(t0, P.X, Q.X, C); g_mul(t1, P.Y, Q.Y, C); g_mul(t2, P.Z, Q.Z, C);
    g_add(t3, P.X, P.Y, C); g_add(t4, Q.X, Q.Y, C); g_mul(t3, t3, t4, C);
    g_add(t4, t0, t1, C);   g_sub(t3, t3, t4, C);   g_add(t4, P.X, P.Z, C);
    g_add(t5, Q.X, Q.Z, C); g_mul(t4, t4, t5, C);   g_add(t5, t0, t2, C);
    g_sub(t4, t4, t5, C);   g_add(t5, P.Y, P.Z, C); g_add(X3, Q.Y, Q.Z, C);
    g_mul(t5, t5, X3, C);   g_add(X3, t1, t2, C);   g_sub(t5, t5, X3, C);
    g_mul(Z3, a, t4, C);    g_mul(X3, b3, t2, C);   g_add(Z3, X3, Z3, C);
    g_sub(X3, t1, Z3, C);   g_add(Z3, t1, Z3, C);   g_mul(Y3, X3, Z3, C);
    g_add(t1, t0, t0, C);   g_add(t1, t1, t0, C);   g_mul(t2, a, t2, C);
    g_mul(t4, b3, t4, C);   g_add(t1, t1, t2, C);   g_sub(t2, t0, t2, C);
    g_mul(t2, a, t2, C);    g_add(t4, t4, t2, C);   g_mul(t0, t1, t4, C);
    g_add(Y3, Y3, t0, C);   g_mul(t0, t5, t4, C);   g_mul(X3, t3, X3, C);
    g_sub(X3, X3, t0, C);   g_mul(t0, t3, t1, C);   g_mul(Z3, t5, Z3, C);
    g_add(Z3, Z3, t0, C);
    R.X = X3; R.Y = Y3; R.Z = Z3;
}

// (x, y) of k G in canonical (non-Montgomery) form; returns all-ones iff k G = O.  Constant-time in k.
template <int N>
__device__ __forceinline__ uint32_t g_mul_base_ct(gfe<N> &x, gfe<N> &y, const uint32_t (&k)[N], const GenCurve<N> &C)
{
    gfe<N> a, b3, t;
    g_set(a, C.a);
    g_set(t, C.b);
    g_add(b3, t, t, C);
    g_add(b3, b3, t, C);
    gproj<N> G, T, U;
#pragma unroll
    for (int i = 0; i < N; ++i) { G.X.v[i] = 0; T.X.v[i] = 0; T.Z.v[i] = 0; }
    g_set(G.Y, C.gy); g_set(G.Z, C.one);
    g_set(T.Y, C.one);
#pragma unroll 1
    for (int i = 32 * N - 1; i >= 0; --i) {
        gp_add_complete(T, T, T, a, b3, C);
        gp_add_complete(U, T, G, a, b3, C);
        const uint32_t m = 0u - ((k[i >> 5] >> (i & 31)) & 1u);
#pragma unroll
        for (int l = 0; l < N; ++l) {
            T.X.v[l] = ct_sel(m, U.X.v[l], T.X.v[l]);
            T.Y.v[l] = ct_sel(m, U.Y.v[l], T.Y.v[l]);
            T.Z.v[l] = ct_sel(m, U.Z.v[l], T.Z.v[l]);
        }
    }
    gfe<N> zi, one;
    g_inv(zi, T.Z, C);                                   // a^(p-2): the exponent is public; Z = 0 gives 0
    g_mul(x, T.X, zi, C);
    g_mul(y, T.Y, zi, C);
#pragma unroll
    for (int i = 0; i < N; ++i) one.v[i] = i == 0;
    g_mul(x, x, one, C);                                 // out of the Montgomery domain
    g_mul(y, y, one, C);
    return ct_is_zero(T.Z.v);
}

// scalars n x 4N octets.  MODE 0: signing (x_R only, every lane computes), 1: bignPubkeyCalc (0 < d < q or ERR_BAD_PRIVKEY,
// refused keys leave zeros), 2: bignKeypairGen (any d; ERR_BAD_PARAMS when d G = O) -- as bign_mulbase_ct_kernel
// (the mode is a launch argument -- public, wavefront-uniform -- so each curve size is ONE kernel of the product library)
template <int N>
__global__ __launch_bounds__(64)
void bign_generic_mulbase_kernel(const uint8_t *__restrict__ scalars, size_t n, uint32_t *__restrict__ codes,
                                 uint8_t *__restrict__ xy_out, GenCurve<N> C, QArg<N> qa, const int MODE)
{
    constexpr int NO = 4 * N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t k[N];
    load_words_bytes(k, scalars + NO * idx);
    uint32_t valid = ~0u;
    if (MODE == 1) {
        valid = ct_in_range_q(k, qa);
        codes[idx] = ct_sel(valid, (uint32_t)ERR_OK, ERR_BAD_PRIVKEY_V);
    }
    gfe<N> x, y;
    const uint32_t inf = g_mul_base_ct(x, y, k, C);
    if (MODE == 2) {
        valid = ~inf;
        codes[idx] = ct_sel(inf, (uint32_t)ERR_BAD_PARAMS, (uint32_t)ERR_OK);
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(xy_out + (size_t)(MODE == 0 ? NO : 2 * NO) * idx);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = x.v[i] & valid;
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) o[N + i] = y.v[i] & valid;
    }
}

// s0 = belt-hash(oid || <x_R> || H)[0 .. l bits), s1 = (k - (s0 + 2^l) d - H) mod q for any odd 2l-bit q
// (bign_sign.c:221-238); Cq = the Montgomery context of q.  Same layout and hash as bign_sign_tail_kernel.
template <int N>
__global__ __launch_bounds__(SIGN_WG)
void bign_generic_sign_tail_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ privkeys,
                                   const uint8_t *__restrict__ rx, uint8_t *__restrict__ ks, size_t n, OidArg oid, GenCurve<N> Cq,
                                   const uint32_t *__restrict__ status, uint8_t *__restrict__ sigs, uint32_t *__restrict__ codes)
{
    constexpr int NO = 4 * N;
    constexpr int ROW = (OID_MAX + 2 * 64 + 31) / 32 * 8 + 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint8_t *s_tab = s_dyn;
    uint32_t *s_rows = reinterpret_cast<uint32_t *>(s_dyn + BeltTabTwo::kBytes);
    BeltTabTwo::fill(s_tab, threadIdx.x, SIGN_WG);
    __syncthreads();
    const BeltTabTwoP T(s_tab);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t st = status[idx];
    const uint32_t ok = ct_eq_small(st >> 1, ST_PENDING >> 1);

    uint32_t xr[N], H[N];
    gfe<N> d, k;
    load_words_bytes(xr, rx + NO * idx);
    load_words_bytes(H, hashes + NO * idx);
    load_words_bytes(d.v, privkeys + NO * idx);
    load_words_bytes(k.v, ks + NO * idx);
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

    // v = s0 + 2^l < 2^(l+1) < q; d < q for the lanes that count (the others are masked out below, but g_mul wants
    // operands below the modulus: a refused d is replaced by 1)
    gfe<N> v, t, r2, Hq;
#pragma unroll
    for (int i = 0; i < N; ++i) v.v[i] = i < N / 2 ? h[i] : i == N / 2 ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) { d.v[i] = ct_sel(ok, d.v[i], i == 0 ? 1u : 0u); k.v[i] = ct_sel(ok, k.v[i], i == 0 ? 1u : 0u); }
    g_set(r2, Cq.r2);
    g_mul(t, v, d, Cq);                              // v d R^-1
    g_mul(t, t, r2, Cq);                             // v d mod q
    // H mod q: one conditional subtraction (zzSubMod's operand, bign_sign.c:231-236), by masks
    {
        uint32_t s[N];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint64_t dd = (uint64_t)H[i] - Cq.p[i] - borrow;
            s[i] = (uint32_t)dd; borrow = (uint32_t)(dd >> 32) & 1u;
        }
        const uint32_t lt = 0u - borrow;
#pragma unroll
        for (int i = 0; i < N; ++i) Hq.v[i] = ct_sel(lt, H[i], s[i]);
    }
    gfe<N> s1;
    g_sub(s1, k, t, Cq);
    g_sub(s1, s1, Hq, Cq);

    uint32_t *so = reinterpret_cast<uint32_t *>(sigs + (NO + NO / 2) * idx);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) so[i] = h[i] & ok;
#pragma unroll
    for (int i = 0; i < N; ++i) so[N / 2 + i] = s1.v[i] & ok;
    codes[idx] = ct_sel(ok, (uint32_t)ERR_OK, st);
}

// ------------------------------------------------------------------ host side ---
namespace {
// little multi-precision helpers on N 32-bit limbs (host; run once per call on the parameter set)
template <int N> bool h_ge(const uint32_t *a, const uint32_t *b)
{
    for (int i = N - 1; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
template <int N> void h_sub(uint32_t *r, const uint32_t *a, const uint32_t *b)
{
    uint64_t borrow = 0;
    for (int i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
        r[i] = (uint32_t)d; borrow = (d >> 32) & 1u;
    }
}
// x <- 2 x mod p (x < p)
template <int N> void h_dbl_mod(uint32_t *x, const uint32_t *p)
{
    const uint32_t top = x[N - 1] >> 31;
    for (int i = N - 1; i > 0; --i) x[i] = (x[i] << 1) | (x[i - 1] >> 31);
    x[0] <<= 1;
    if (top || h_ge<N>(x, p)) h_sub<N>(x, x, p);
}
template <int N> void h_load(uint32_t *w, const octet *src)
{
    for (int i = 0; i < N; ++i)
        w[i] = (uint32_t)src[4 * i] | (uint32_t)src[4 * i + 1] << 8 | (uint32_t)src[4 * i + 2] << 16 | (uint32_t)src[4 * i + 3] << 24;
}
// x R mod p by 32 N modular doublings
template <int N> void h_to_mont(uint32_t *r, const uint32_t *x, const uint32_t *p)
{
    for (int i = 0; i < N; ++i) r[i] = x[i];
    for (int i = 0; i < 32 * N; ++i) h_dbl_mod<N>(r, p);
}

// bignEcCreate's checks beyond bignParamsCheck (bign_ec.c:64-70: gfpCreate, ecpCreateJ, ecGroupCreate): a, b, yG < p
template <int N>
err_t make_curve(GenCurve<N> &C, const bign_params *params)
{
    uint32_t a[N], b[N], gy[N];
    h_load<N>(C.p, params->p);
    h_load<N>(C.q, params->q);
    h_load<N>(a, params->a);
    h_load<N>(b, params->b);
    h_load<N>(gy, params->yG);
    if (h_ge<N>(a, C.p) || h_ge<N>(b, C.p) || h_ge<N>(gy, C.p)) return ERR_BAD_PARAMS;
    // R mod p = 2^(32N) - p (p has its top bit set, so one subtraction)
    uint32_t zero[N];
    for (int i = 0; i < N; ++i) zero[i] = 0;
    h_sub<N>(C.one, zero, C.p);
    h_to_mont<N>(C.r2, C.one, C.p);
    h_to_mont<N>(C.a, a, C.p);
    h_to_mont<N>(C.b, b, C.p);
    h_to_mont<N>(C.gy, gy, C.p);
    uint32_t two[N];
    for (int i = 0; i < N; ++i) two[i] = i == 0 ? 2u : 0u;
    h_sub<N>(C.pm2, C.p, two);
    uint32_t x = 1;                                  // p^-1 mod 2^32 by Newton iteration
    for (int i = 0; i < 6; ++i) x *= 2u - C.p[0] * x;
    C.n0 = 0u - x;
    return ERR_OK;
}

template <int N>
err_t verify_generic_t(const bign_params *params, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                       const void *d_sigs, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st)
{
    GenCurve<N> C;
    err_t code = make_curve<N>(C, params);
    if (code != ERR_OK) return code;
    VerifyScratch S;
    code = bign_scratch<N>(st, n, S);
    if (code != ERR_OK) return code;
    OidArg oid;
    code = make_oid_arg(oid, oid_der, oid_len, st);
    if (code != ERR_OK) return code;
    const unsigned g64 = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(bign_generic_verify_kernel<N>, dim3(g64), dim3(64), 0, st, (const uint8_t *)d_hashes,
                       (const uint8_t *)d_sigs, (const uint8_t *)d_pubkeys, n, S, C);
    constexpr size_t row_bytes = (2 * N + 1) * 4;
    hipLaunchKernelGGL((bign_tail_kernel<N, BeltTabSmall, 64>), dim3(g64), dim3(64), BeltTabSmall::kBytes + 64 * row_bytes, st,
                       (const uint8_t *)d_hashes, (const uint8_t *)d_sigs, n, S, oid, (uint32_t *)d_codes);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// Montgomery context of an odd 2l-bit modulus m (the fields of GenCurve that g_mul / g_add / g_sub read)
template <int N>
void make_mod(GenCurve<N> &C, const octet *m)
{
    memset(&C, 0, sizeof C);
    h_load<N>(C.p, m);
    uint32_t zero[N];
    for (int i = 0; i < N; ++i) zero[i] = 0;
    h_sub<N>(C.one, zero, C.p);
    h_to_mont<N>(C.r2, C.one, C.p);
    uint32_t x = 1;
    for (int i = 0; i < 6; ++i) x *= 2u - C.p[0] * x;
    C.n0 = 0u - x;
}

template <int N>
err_t pubkey_calc_generic_t(const bign_params *params, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes,
                            hipStream_t st)
{
    GenCurve<N> C;
    err_t code = make_curve<N>(C, params);
    if (code != ERR_OK) return code;
    QArg<N> qa;
    memset(&qa, 0, sizeof qa);
    h_load<N>(qa.q, params->q);
    const unsigned g64 = (unsigned)((n + 63) / 64);
    if (!keygen)
        hipLaunchKernelGGL((bign_generic_mulbase_kernel<N>), dim3(g64), dim3(64), 0, st, (const uint8_t *)d_privkeys, n,
                           (uint32_t *)d_codes, (uint8_t *)d_pubkeys, C, qa, 1);
    else
        hipLaunchKernelGGL((bign_generic_mulbase_kernel<N>), dim3(g64), dim3(64), 0, st, (const uint8_t *)d_privkeys, n,
                           (uint32_t *)d_codes, (uint8_t *)d_pubkeys, C, qa, 2);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// modes as launch_bign_sign_t (bign_sign_kernels.hip): 0 deterministic with t, 1 one-time keys supplied, 2 theta supplied
template <int N>
err_t sign_generic_t(const bign_params *params, int mode, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                     const void *d_privkeys, const void *d_aux, size_t t_len, int t_shared, size_t n, void *d_sigs, void *d_codes,
                     hipStream_t st)
{
    if (n == 0) return ERR_OK;
    if (mode == 0 && d_aux && t_len > (size_t)SIGN_T_MAX) return ERR_NOT_IMPLEMENTED;     // the host hashes longer t (mode 2)
    GenCurve<N> C, Cq;
    err_t code = make_curve<N>(C, params);
    if (code != ERR_OK) return code;
    make_mod<N>(Cq, params->q);
    QArg<N> qa;
    memset(&qa, 0, sizeof qa);
    h_load<N>(qa.q, params->q);
    OidArg oa;
    code = make_oid_arg(oa, oid_der, oid_len, st);
    if (code != ERR_OK) return code;
    SignScratch S;
    code = sign_scratch<N>(st, n, S);
    if (code != ERR_OK) return code;
    const unsigned grid = (unsigned)((n + SIGN_WG - 1) / SIGN_WG);
    if (mode == 0 || mode == 2) {
        constexpr int ROW = (OID_MAX + 64 + SIGN_T_MAX + 31) / 32 * 8 + 1;
        const size_t lds = BeltTabTwo::kBytes + (size_t)SIGN_WG * ROW * 4;
        B2H_TRY(dyn_lds_once((const void *)bign_sign_nonce_kernel<N>, lds));
        const uint8_t *tp = mode == 0 ? (const uint8_t *)d_aux : nullptr;
        hipLaunchKernelGGL(bign_sign_nonce_kernel<N>, dim3(grid), dim3(SIGN_WG), lds, st, (const uint8_t *)d_hashes,
                           (const uint8_t *)d_privkeys, tp, (uint32_t)(tp ? t_len : 0), (uint32_t)(t_shared ? 0 : t_len),
                           mode == 2 ? (const uint8_t *)d_aux : nullptr, n, oa, qa, S.status, S.k, (uint32_t)ROW);
    } else {
        hipLaunchKernelGGL(bign_sign_kcheck_kernel<N>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_privkeys, (const uint8_t *)d_aux, n, qa, S.status);
        B2H_TRY(hipMemcpyAsync(S.k, d_aux, n * 4 * N, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL((bign_generic_mulbase_kernel<N>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const uint8_t *)S.k, n,
                       (uint32_t *)nullptr, S.rx, C, qa, 0);
    {
        constexpr int ROW = (OID_MAX + 2 * 64 + 31) / 32 * 8 + 1;
        const size_t lds = BeltTabTwo::kBytes + (size_t)SIGN_WG * ROW * 4;
        B2H_TRY(dyn_lds_once((const void *)bign_generic_sign_tail_kernel<N>, lds));
        hipLaunchKernelGGL(bign_generic_sign_tail_kernel<N>, dim3(grid), dim3(SIGN_WG), lds, st, (const uint8_t *)d_hashes,
                           (const uint8_t *)d_privkeys, (const uint8_t *)S.rx, S.k, n, oa, Cq, (const uint32_t *)S.status,
                           (uint8_t *)d_sigs, (uint32_t *)d_codes);
    }
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}
}  // namespace

// params: already through bignParamsCheck's tests (capi.hip params_check), l in {128, 192, 256}, not a standard set
err_t launch_bign_verify_generic(const bign_params *params, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                                 const void *d_sigs, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st)
{
    if (params->l == 128) return verify_generic_t<8>(params, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, st);
    if (params->l == 192) return verify_generic_t<12>(params, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, st);
    if (params->l == 256) return verify_generic_t<16>(params, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, st);
    return ERR_BAD_PARAMS;
}

namespace {
template <int N>
err_t pubkey_val_generic_t(const bign_params *params, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st)
{
    GenCurve<N> C;
    const err_t code = make_curve<N>(C, params);
    if (code != ERR_OK) return code;
    if (n == 0) return ERR_OK;
    hipLaunchKernelGGL(bign_generic_pubkey_val_kernel<N>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st,
                       (const uint8_t *)d_pubkeys, n, (uint32_t *)d_codes, C);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}
}  // namespace

// what bignEcCreate adds to bignParamsCheck (bign_ec.c:64-70): ERR_BAD_PARAMS unless a, b, yG < p.  Host only.
err_t bign_generic_check(const bign_params *params)
{
    if (params->l == 128) { GenCurve<8> C; return make_curve<8>(C, params); }
    if (params->l == 192) { GenCurve<12> C; return make_curve<12>(C, params); }
    if (params->l == 256) { GenCurve<16> C; return make_curve<16>(C, params); }
    return ERR_BAD_PARAMS;
}

err_t launch_bign_pubkey_val_generic(const bign_params *params, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st)
{
    if (params->l == 128) return pubkey_val_generic_t<8>(params, d_pubkeys, n, d_codes, st);
    if (params->l == 192) return pubkey_val_generic_t<12>(params, d_pubkeys, n, d_codes, st);
    if (params->l == 256) return pubkey_val_generic_t<16>(params, d_pubkeys, n, d_codes, st);
    return ERR_BAD_PARAMS;
}

// the signing side on a non-standard parameter set (params through params_check2; l in {128, 192, 256})
err_t launch_bign_pubkey_calc_generic(const bign_params *params, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys,
                                      void *d_codes, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    if (params->l == 128) return pubkey_calc_generic_t<8>(params, keygen, d_privkeys, n, d_pubkeys, d_codes, st);
    if (params->l == 192) return pubkey_calc_generic_t<12>(params, keygen, d_privkeys, n, d_pubkeys, d_codes, st);
    if (params->l == 256) return pubkey_calc_generic_t<16>(params, keygen, d_privkeys, n, d_pubkeys, d_codes, st);
    return ERR_BAD_PARAMS;
}
err_t launch_bign_sign_generic(const bign_params *params, int mode, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                               const void *d_privkeys, const void *d_aux, size_t t_len, int t_shared, size_t n, void *d_sigs,
                               void *d_codes, hipStream_t st)
{
    if (params->l == 128) return sign_generic_t<8>(params, mode, oid_der, oid_len, d_hashes, d_privkeys, d_aux, t_len, t_shared, n, d_sigs, d_codes, st);
    if (params->l == 192) return sign_generic_t<12>(params, mode, oid_der, oid_len, d_hashes, d_privkeys, d_aux, t_len, t_shared, n, d_sigs, d_codes, st);
    if (params->l == 256) return sign_generic_t<16>(params, mode, oid_der, oid_len, d_hashes, d_privkeys, d_aux, t_len, t_shared, n, d_sigs, d_codes, st);
    return ERR_BAD_PARAMS;
}

}  // namespace bee2hip
