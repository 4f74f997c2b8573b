// bign_kernels.hip -- batched bign signature verification on the three standard bign curves
// (gfx950): bign-curve256v1 (l = 128, the graded configuration), 384v1 (l = 192), 512v1 (l = 256).
//
// H3 of SURVEY.md 8a (+ 8f-4).  Replaces n calls of bignVerify / bign128Verify / bign192Verify /
// bign256Verify (src/crypto/bign/bign_sign.c:268-361, bign128.c:177-185, bign192.c, bign256.c).
// One lane per signature; per-signature result is the bee2 err_t the reference returns.
//
// The reference computes R = s1' G + (s0 + 2^l) Q with interleaved width-5/6 NAF
// (ecAddMulA, src/math/ec.c:1183-1273): data-dependent branching, 2l+1 doublings.  For a public
// key ON the curve any correct algorithm yields the same affine R, so the GPU uses a
// wavefront-friendly schedule instead (N = number of 32-bit limbs = l/16).  (Neither bignVerify
// nor this code checks that Q is on the curve; for an off-curve Q the a = -3 formulas never see
// b and "u G + v Q" depends on the order of additions, so R differs between the two -- both then
// return ERR_BAD_SIG because the hash of x_R matches s0 with probability 2^-l only.  Parity on
// off-curve keys is therefore probabilistic, not structural; callers run bignPubkeyVal first,
// as bee2cmd does.)
//   * G part: fixed-base comb, no doublings: 2N windows x 16 bits, 2N x 65535 affine points
//     (64 / 144 / 256 MiB for l = 128 / 192 / 256, built once per device and curve in 3 / 10 /
//     22 ms: bign_gtable_kernel makes an 8-bit seed table by double-and-add, bign_gtable16_kernel
//     combines it) -> 16 / 24 / 32 mixed additions.
//   * Q part: signed radix-16 digits of the (l+1)-bit scalar (uniform 4 doublings + 1
//     mixed addition per digit, 4N+1 digits), per-signature AFFINE table 1Q..8Q kept in an HBM
//     scratch laid out [entry][limb][signature] so table reads coalesce across the wavefront.
//   * inversions (table normalisation, x_R = X / Z^2) by division steps (bign_dev.hpp,
//     fe_inv_safegcd) and shared between signatures (Montgomery's trick).
// Exceptional cases of the addition law (operand O, P = +-Q) cannot occur for honest
// inputs; lanes that hit one are flagged and recomputed by bign_slow_kernel with the
// complete (branchy) formulas, so verdicts are exact for every on-curve key.
//
// Kernels per batch (same stream): [points ->] prep -> main -> slow -> inv -> tail; on the 256-bit curve the batch size
// picks who walks the scalar multiplication: bign_quad29_kernel (= prep + main; a quad of lanes per signature up to
// 2^14 signatures, a pair up to 2^15), prep + bign_main29_kernel (29-bit limbs, up to 2^18 since round 6), prep + bign_main_kernel.
//   prep : range checks (bign_sign.c:306-318), u = s1 + H mod q (:320-327),
//          v = s0 + 2^l (:329-330), affine Q table (on the wider curves the first half of this,
//          everything up to the Jacobian table points, is a kernel of its own: points)
//   main : the double-scalar multiplication, leaves R as (X, Z)
//   slow : flagged lanes only (complete formulas, own inversion)
//   inv  : x_R = X / Z^2, one inversion per K signatures
//   tail : belt-hash(oid || x_R || H) == s0 ? (bign_sign.c:337-343)
// HBM traffic is irrelevant here (148 B of input per ~5e5 VALU ops at l = 128): the bound is
// the integer multiplier rate.
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>
#include "belt_dev.hpp"
#include "bign_dev.hpp"
#include "bign_fe29.hpp"
#include "bign_quad29.hpp"
#include "common.hpp"
#include "host_bign.hpp"          // host arithmetic of the one-signer path: the points 2^(8 w) Q (verification: no secrets)
#include "bign_curves.inc"

namespace bee2hip {

// status word per signature while the batch is in flight
constexpr uint32_t ST_PENDING = 0xFFFFFFFFu;     // fast path result in rx[]
constexpr uint32_t ST_SLOW = 0xFFFFFFFEu;        // needs the complete slow path
// anything else: a final bee2 err_t

// q (group order) and yG per curve, STB 34.101.45 annex B (bign_params.c:34-178), LE limbs
__constant__ uint32_t c_q8[8] = BIGN128_Q_LIMBS;
__constant__ uint32_t c_yG8[8] = BIGN128_YG_LIMBS;
__constant__ uint32_t c_q12[12] = BIGN192_Q_LIMBS;
__constant__ uint32_t c_yG12[12] = BIGN192_YG_LIMBS;
__constant__ uint32_t c_q16[16] = BIGN256_Q_LIMBS;
__constant__ uint32_t c_yG16[16] = BIGN256_YG_LIMBS;
__constant__ uint32_t c_b8[8] = BIGN128_B_LIMBS;
__constant__ uint32_t c_b12[12] = BIGN192_B_LIMBS;
__constant__ uint32_t c_b16[16] = BIGN256_B_LIMBS;
template <int N> __device__ __forceinline__ const uint32_t *curve_b() { return N == 8 ? c_b8 : N == 12 ? c_b12 : c_b16; }
template <int N> __device__ __forceinline__ const uint32_t *curve_q() { return N == 8 ? c_q8 : N == 12 ? c_q12 : c_q16; }
template <int N> __device__ __forceinline__ const uint32_t *curve_yG() { return N == 8 ? c_yG8 : N == 12 ? c_yG12 : c_yG16; }

// comb geometry: window width W bits, 32N/W windows, 2^W entries (entry 0 = neutral, unused),
// each entry an affine point of 2N limbs.  16 bits on every curve: the wider curves ran the 8-bit seed table
// directly at first (48 / 64 additions instead of 24 / 32, a fifth of their work) -- +10 % / +11 % with the big one.
template <int N> struct Comb { static constexpr int W = 16; };
constexpr int GT8_ENTRIES = 256;

struct VerifyScratch {          // all arrays are [..][n_pad] (signature index fastest)
    uint32_t *status;           // [n_pad]
    uint32_t *u;                // [N][n_pad]      scalar of G; after main / slow: Z of R
    uint32_t *w;                // [N/2+1][n_pad]  v + 0x888..8 (4N+1 nibbles): digit_i = nib_i - 8
    uint32_t *qtab;             // [8][2N][n_pad]  1Q..8Q, affine once bign_prep_kernel is through
    uint32_t *qz;               // [14][N][n_pad]  prep: Z of 2Q..8Q and the running products; inv: prefixes
    uint32_t *rx;               // [N][n_pad]      X of R, then (bign_inv_kernel) canonical x_R
    size_t n_pad;
};

// ------------------------------------------------------------------ loaders ---
template <int N>
__device__ __forceinline__ void load_fe_bytes(feT<N> &r, const uint8_t *p)
{
    // 4N little-endian octets -> N limbs (wwFrom, src/math/ww.c); p is 16-byte aligned
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
        const uint4 a = *reinterpret_cast<const uint4 *>(p + 16 * k);
        r.v[4 * k] = a.x; r.v[4 * k + 1] = a.y; r.v[4 * k + 2] = a.z; r.v[4 * k + 3] = a.w;
    }
}
// same from a pointer that is only 4-byte aligned (the s1 half of a 72-byte signature)
template <int N>
__device__ __forceinline__ void load_fe_words(feT<N> &r, const uint8_t *p)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = w[i];
}
template <int N>
__device__ __forceinline__ void store_soa(uint32_t *base, size_t n_pad, size_t idx, const feT<N> &a)
{
#pragma unroll
    for (int l = 0; l < N; ++l) base[(size_t)l * n_pad + idx] = a.v[l];
}
template <int N>
__device__ __forceinline__ void load_soa(feT<N> &a, const uint32_t *base, size_t n_pad, size_t idx)
{
#pragma unroll
    for (int l = 0; l < N; ++l) a.v[l] = base[(size_t)l * n_pad + idx];
}
template <int N>
__device__ __forceinline__ void store_qxy(const VerifyScratch &S, int e, size_t idx, const feT<N> &x, const feT<N> &y)
{
    uint32_t *b = S.qtab + (size_t)e * 2 * N * S.n_pad;
    store_soa(b, S.n_pad, idx, x);
    store_soa(b + (size_t)N * S.n_pad, S.n_pad, idx, y);
}
template <int N>
__device__ __forceinline__ void load_qaff(affT<N> &P, const VerifyScratch &S, int e, size_t idx)
{
    const uint32_t *b = S.qtab + (size_t)e * 2 * N * S.n_pad;
    load_soa(P.x, b, S.n_pad, idx);
    load_soa(P.y, b + (size_t)N * S.n_pad, S.n_pad, idx);
}
template <int N>
__device__ __forceinline__ void load_aff(affT<N> &E, const uint4 *e)
{
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
        const uint4 a = e[k], b = e[N / 4 + k];
        E.x.v[4 * k] = a.x; E.x.v[4 * k + 1] = a.y; E.x.v[4 * k + 2] = a.z; E.x.v[4 * k + 3] = a.w;
        E.y.v[4 * k] = b.x; E.y.v[4 * k + 1] = b.y; E.y.v[4 * k + 2] = b.z; E.y.v[4 * k + 3] = b.w;
    }
}
template <int N>
__device__ __forceinline__ void store_aff(uint4 *e, const feT<N> &x, const feT<N> &y)
{
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
        e[k] = make_uint4(x.v[4 * k], x.v[4 * k + 1], x.v[4 * k + 2], x.v[4 * k + 3]);
        e[N / 4 + k] = make_uint4(y.v[4 * k], y.v[4 * k + 1], y.v[4 * k + 2], y.v[4 * k + 3]);
    }
}
// (X : Y : Z) -> canonical affine (x, y); Z != 0
template <int N>
__device__ __forceinline__ void to_affine(feT<N> &x, feT<N> &y, const jacT<N> &T)
{
    feT<N> zi = fe_inv_checked(T.Z), zi2;
    fe_sqr(zi2, zi);
    fe_mul(x, T.X, zi2);
    fe_mul(zi2, zi2, zi);
    fe_mul(y, T.Y, zi2);
    fe_canon(x, x);
    fe_canon(y, y);
}

// --------------------------------------------------------------------- prep ---
// one signature: range checks (status written on failure), Q, and the scalars u and w, which also go to the scratch
// (the slow path reads u from there).  `write` = false for the lanes of a quad that only need the values.
template <int N>
__device__ __forceinline__ bool prep_scalars(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                                             const uint8_t *__restrict__ pubkeys, size_t idx, const VerifyScratch &S,
                                             bool write, affT<N> &Q, feT<N> &u, uint32_t (&w)[N / 2 + 1],
                                             size_t pk_stride = 8 * N)
{
    constexpr int NO = 4 * N;                       // octets per field element

    // pk_stride: octets between public keys (2 NO; 0 = ONE key for the whole batch, bign_onekey_kernel)
    load_fe_bytes(Q.x, pubkeys + pk_stride * idx);
    load_fe_bytes(Q.y, pubkeys + pk_stride * idx + NO);
    feT<N> s1, H;
    const uint8_t *sig = sigs + (NO + NO / 2) * idx;   // s0 (NO/2 octets) || s1 (NO octets)
    load_fe_words(s1, sig + NO / 2);
    load_fe_bytes(H, hashes + NO * idx);

    uint32_t q[N], P[N];
    const uint32_t *cq = curve_q<N>();
#pragma unroll
    for (int i = 0; i < N; ++i) { q[i] = cq[i]; P[i] = 0xFFFFFFFFu; }
    P[0] = 0u - CurveC<N>::C;

    // qrFrom rejects coordinates >= p (bign_sign.c:306-311); there is no on-curve check
    if (limbs_ge(Q.x.v, P) || limbs_ge(Q.y.v, P)) { if (write) S.status[idx] = ERR_BAD_PUBKEY; return false; }
    // s1 >= q (bign_sign.c:313-318)
    if (limbs_ge(s1.v, q)) { if (write) S.status[idx] = ERR_BAD_SIG; return false; }

    // H <- H - q if H >= q ; u <- (s1 + H) mod q   (bign_sign.c:320-327)
    {
        uint32_t t[N];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint64_t d = (uint64_t)H.v[i] - q[i] - borrow;
            t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) H.v[i] = borrow ? H.v[i] : t[i];
        uint64_t c = 0;
        uint32_t s[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { c += (uint64_t)s1.v[i] + H.v[i]; s[i] = (uint32_t)c; c >>= 32; }
        const uint32_t carry = (uint32_t)c;
        borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint64_t d = (uint64_t)s[i] - q[i] - borrow;
            t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
        }
        const bool ge = carry || !borrow;            // s1 + H >= q
#pragma unroll
        for (int i = 0; i < N; ++i) u.v[i] = ge ? t[i] : s[i];
        if (write) store_soa(S.u, S.n_pad, idx, u);
    }
    // v = s0 + 2^l ; w = v + 0x8888...8 (4N+1 nibbles) so that digit_i = nibble_i(w) - 8
    {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i <= N / 2; ++i) {
            const uint32_t vi = i < N / 2 ? *reinterpret_cast<const uint32_t *>(sig + 4 * i) : 1u;
            c += (uint64_t)vi + (i < N / 2 ? 0x88888888u : 0x8u);
            w[i] = (uint32_t)c;
            if (write) S.w[(size_t)i * S.n_pad + idx] = (uint32_t)c;
            c >>= 32;
        }
    }
    return true;
}

// one signature: prep_scalars, then 1Q..8Q (1Q affine, 2Q..8Q Jacobian: X, Y in the table rows, Z aside).  Sets
// the status; returns true when the table is to be normalised (no exceptional case).
// order in which prep_points produces the entries 2Q..8Q (entry e = (e + 1) Q): slot k of the running product belongs to entry prep_ord(k) = 1, 2, 3, 5, 6, 4, 7
// (nibble k of a literal: an array indexed by a loop counter would be parked in LDS / scratch)
__device__ __forceinline__ constexpr int prep_ord(int k) { return (int)((0x7465321u >> (4 * k)) & 15u); }
// ACC: acc is the running product of the Z of the points as they come out -- slot k of Cs gets the product BEFORE entry
// prep_ord(k), then acc takes its Z.  Round 5: the normalisation's first pass used to read every Z back from memory, one dependent
// load in front of every multiplication of a latency-bound kernel (bign_prep_kernel: 5.7 cycles per VALU instruction, valu_busy 0.68).
template <int N, class OPS = VtOps, bool ACC = false>
__device__ __forceinline__ bool prep_points(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                                            const uint8_t *__restrict__ pubkeys, size_t idx, const VerifyScratch &S,
                                            feT<N> &acc, uint32_t *Cs)
{
    affT<N> Q;
    {
        feT<N> u;
        uint32_t w[N / 2 + 1];
        if (!prep_scalars<N>(hashes, sigs, pubkeys, idx, S, true, Q, u, w)) return false;
    }
    // table 1Q..8Q: 1Q as given, 2Q..8Q Jacobian for now (bign_prep_kernel normalises them).  Any exceptional
    // case -> slow path.
    bool ok = true;
    int slot = 0;
    const auto put = [&](int e, const jacT<N> &P) {      // entry e = (e+1) Q: X, Y in place, Z aside
        store_qxy(S, e, idx, P.X, P.Y);
        store_soa(S.qz + (size_t)(e - 1) * N * S.n_pad, S.n_pad, idx, P.Z);
        ok &= !fe_is_zero(P.Z);
        if constexpr (ACC) {                                 // (acc by reference and a compile-time switch: through a pointer it was parked in LDS)
            store_soa(Cs + (size_t)slot * N * S.n_pad, S.n_pad, idx, acc);
            fe_mul<1, OPS>(acc, acc, P.Z);
            ++slot;
        }
    };
    // Two chains, two points in registers: A = 2Q -> 4Q -> 8Q and T = 3Q -> 6Q -> 7Q, then 5Q = 4Q + Q from A.
    jacT<N> A, T;
    store_qxy(S, 0, idx, Q.x, Q.y);
    A.X = Q.x; A.Y = Q.y; fe_set_one(A.Z);
    jac_dbl<N, OPS>(A);                           put(1, A);      // 2Q
    T = A;   ok &= jac_madd<N, OPS>(T, Q);        put(2, T);      // 3Q
    jac_dbl<N, OPS>(A);                           put(3, A);      // 4Q
    jac_dbl<N, OPS>(T);                           put(5, T);      // 6Q
    ok &= jac_madd<N, OPS>(T, Q);                 put(6, T);      // 7Q
    T = A;   ok &= jac_madd<N, OPS>(T, Q);        put(4, T);      // 5Q
    jac_dbl<N, OPS>(A);                           put(7, A);      // 8Q
    S.status[idx] = ok ? ST_PENDING : ST_SLOW;
    return ok;
}

template <int N, class OPS = VtOps>
__global__ __launch_bounds__(256, 2)
void bign_points_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                        const uint8_t *__restrict__ pubkeys, size_t n, VerifyScratch S)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    feT<N> none;
    if (idx < n) prep_points<N, OPS, false>(hashes, sigs, pubkeys, idx, S, none, nullptr);
}

// The table is made AFFINE so that the main loop adds with the mixed formula (8M + 3S instead of 12M + 4S, 31
// times per signature): the Jacobian points 2Q..8Q are normalised with a shared inversion (Montgomery's trick
// over their Z, inversion by division steps).  A lane takes SP signatures (j, j + lanes, ...) and inverts once
// for all of them: at 2 wavefronts per SIMD (196 VGPRs) 2^18 signatures are two rounds of wavefronts anyway, so
// SP = 2 halves the inversions at no loss of parallelism.  Signatures whose table hit an exceptional case go to
// the slow path and stay out of the product.
// (two wavefronts per SIMD on every curve: 196 VGPRs on the 256-bit one as it comes; the wider ones, 311 / 409
// VGPRs unbounded, are held to 256 -- +3 % on their whole pipelines)
template <int N, class OPS = VtOps>
__global__ __launch_bounds__(256, (N == 8 ? 1 : 2))
void bign_prep_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                      const uint8_t *__restrict__ pubkeys, size_t n, size_t lanes, int SP, VerifyScratch S)
{
    // wider curves: the points come from bign_points_kernel -- together the two halves need 311 / 409 VGPRs,
    // apart the normalising half fits 120 / 153 without spills (split: 970 -> 625 us and 1670 -> 1170 us at
    // 2^18 signatures; on the 256-bit curve, 173 VGPRs in one piece, splitting costs 2 %)
    constexpr bool SPLIT = N != 8;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lanes) return;
    uint32_t *Zs = S.qz, *Cs = S.qz + (size_t)7 * N * S.n_pad;     // Z of 2Q..8Q; product of all Z before it
    // (round 5) Every load of this kernel sat directly in front of the multiplication that needs it, at two wavefronts per SIMD: the
    // loops below ask for the NEXT step's operands before they multiply, and on the 256-bit curve the first pass takes the Z of a
    // point while it is still in registers (prep_points' running product) -- slots in prep_ord order there, in entry order when the
    // points come from bign_points_kernel.
    feT<N> acc, z, zi, zi2, v;
    fe_set_one(acc);
    unsigned todo = 0;
#pragma unroll 1
    for (int sp = 0; sp < SP; ++sp) {
        const size_t idx = (size_t)sp * lanes + j;
        if (idx >= n) break;
        if (SPLIT) {
            if (S.status[idx] != ST_PENDING) continue;
#pragma unroll 1
            for (int k = 0; k < 7; ++k) {
                store_soa(Cs + (size_t)k * N * S.n_pad, S.n_pad, idx, acc);
                load_soa(z, Zs + (size_t)k * N * S.n_pad, S.n_pad, idx);
                fe_mul<1, OPS>(acc, acc, z);
            }
        } else {
            const feT<N> before = acc;
            if (!prep_points<N, OPS, true>(hashes, sigs, pubkeys, idx, S, acc, Cs)) { acc = before; continue; }   // (slow path: out of the product)
        }
        todo |= 1u << sp;
    }
    if (!todo) return;
    feT<N> inv = fe_inv_checked(acc);
#pragma unroll 1
    for (int sp = SP - 1; sp >= 0; --sp) {
        if (!((todo >> sp) & 1u)) continue;
        const size_t idx = (size_t)sp * lanes + j;
        if constexpr (N != 8) {
            // the wider curves hold this kernel to 256 VGPRs already: no second operand set there, the loads stay in front of their use
            // (asking ahead measured +1..+4 % WORSE on their pipelines at 2^16..2^18, profiles/r05_ask_ahead_ab.txt)
#pragma unroll 1
            for (int k = 6; k >= 0; --k) {
                load_soa(v, Cs + (size_t)k * N * S.n_pad, S.n_pad, idx);
                fe_mul<1, OPS>(zi, inv, v);                           // 1 / Z_k
                load_soa(z, Zs + (size_t)k * N * S.n_pad, S.n_pad, idx);
                fe_mul<1, OPS>(inv, inv, z);                          // 1 / (product before Z_k)
                uint32_t *b = S.qtab + (size_t)(k + 1) * 2 * N * S.n_pad;
                fe_sqr<1, OPS>(zi2, zi);
                load_soa(v, b, S.n_pad, idx);
                fe_mul<1, OPS>(v, v, zi2);
                store_soa(b, S.n_pad, idx, v);                // x = X / Z^2
                fe_mul<1, OPS>(zi2, zi2, zi);
                load_soa(v, b + (size_t)N * S.n_pad, S.n_pad, idx);
                fe_mul<1, OPS>(v, v, zi2);
                store_soa(b + (size_t)N * S.n_pad, S.n_pad, idx, v);   // y = Y / Z^3
            }
        } else {
        feT<N> cn, zn, xn, yn, x, y;
        const auto ask = [&](int k) {                        // the operands of slot k: product before it, its Z, its X and Y
            const int e = prep_ord(k);
            const uint32_t *b = S.qtab + (size_t)e * 2 * N * S.n_pad;
            load_soa(cn, Cs + (size_t)k * N * S.n_pad, S.n_pad, idx);
            load_soa(zn, Zs + (size_t)(e - 1) * N * S.n_pad, S.n_pad, idx);
            load_soa(xn, b, S.n_pad, idx);
            load_soa(yn, b + (size_t)N * S.n_pad, S.n_pad, idx);
        };
        ask(6);
#pragma unroll 1
        for (int k = 6; k >= 0; --k) {
            v = cn; z = zn; x = xn; y = yn;
            if (k > 0) ask(k - 1);
            fe_mul<1, OPS>(zi, inv, v);                           // 1 / Z_k
            fe_mul<1, OPS>(inv, inv, z);                          // 1 / (product before Z_k)
            uint32_t *b = S.qtab + (size_t)prep_ord(k) * 2 * N * S.n_pad;
            fe_sqr<1, OPS>(zi2, zi);
            fe_mul<1, OPS>(x, x, zi2);
            store_soa(b, S.n_pad, idx, x);                        // x = X / Z^2
            fe_mul<1, OPS>(zi2, zi2, zi);
            fe_mul<1, OPS>(y, y, zi2);
            store_soa(b + (size_t)N * S.n_pad, S.n_pad, idx, y);  // y = Y / Z^3
        }
        }
    }
}


// --------------------------------------------------------------------- main ---
// (second bound: wavefronts per SIMD the register allocation must leave room for.  The 256-bit kernel sits a few
// registers above the 128 that four wavefronts allow unless told so; the 512-bit one took 292 VGPRs, i.e. ONE
// wavefront per SIMD, which issues at under half a SIMD's rate -- held to 256 (36 spills) it runs 1.5x as fast.
// The 384-bit kernel has two wavefronts at 224 VGPRs; forcing three costs 92 spills and 12 %.)
template <int N, class OPS = VtOps>
__global__ __launch_bounds__(256, (N == 8 ? 4 : N == 12 ? 1 : 2))
void bign_main_kernel(size_t n, VerifyScratch S, const uint4 *__restrict__ gtab)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (S.status[idx] != ST_PENDING) return;
    constexpr int NW = N / 2 + 1;                   // limbs of w
    constexpr int W = Comb<N>::W;

    uint32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = S.w[(size_t)i * S.n_pad + idx];
    bool ok = true;

    // top digit d_{4N} = nibble_{4N}(w) - 8 is 1 or 2
    jacT<N> T;
    {
        affT<N> E;
        load_qaff(E, S, (int)(w[NW - 1] & 15u) - 9, idx);
        T.X = E.x; T.Y = E.y; fe_set_one(T.Z);
    }

#pragma unroll 1
    for (int i = 4 * N - 1; i >= 0; --i) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) jac_dbl<N, OPS>(T);
        const int d = (int)(w[NW - 2] >> 28) - 8;       // next digit, in [-8, 7]
#pragma unroll
        for (int l = NW - 2; l > 0; --l) w[l] = (w[l] << 4) | (w[l - 1] >> 28);
        w[0] <<= 4;
        if (d != 0) {
            affT<N> E;
            load_qaff(E, S, (d < 0 ? -d : d) - 1, idx);
            if (d < 0) fe_neg<OPS>(E.y, E.y);
            ok &= jac_madd<N, OPS>(T, E);
        }
    }
    // + u G : comb over the W-bit windows of u (mixed additions only)
    feT<N> u;
    load_soa(u, S.u, S.n_pad, idx);
#pragma unroll 1
    for (int win = 0; win < 32 * N / W; ++win) {
        const uint32_t b = u.v[0] & ((1u << W) - 1u);
#pragma unroll
        for (int l = 0; l < N - 1; ++l) u.v[l] = (u.v[l] >> W) | (u.v[l + 1] << (32 - W));
        u.v[N - 1] >>= W;
        if (b != 0) {
            affT<N> E;
            load_aff(E, gtab + ((size_t)win * (1u << W) + b) * (N / 2));
            ok &= jac_madd<N, OPS>(T, E);
        }
    }
    ok &= !fe_is_zero(T.Z);
    if (!ok) { S.status[idx] = ST_SLOW; return; }
    // x_R = X / Z^2 (ecpToAJ, ecp_j.c:104-133) is finished by bign_inv_kernel, which shares one
    // inversion between several signatures; X goes to rx[], Z takes the place of u (used up)
    store_soa(S.rx, S.n_pad, idx, T.X);
    store_soa(S.u, S.n_pad, idx, T.Z);
}

// ------------------------------------------------------- main, small batches ---
// The same double-scalar multiplication on the signed 29-bit limbs of bign_fe29.hpp, for batches that leave a
// wavefront alone on its SIMD (<= 2^16 signatures: there a kernel costs its instruction count) -- and, since round 6 made its
// multiplication one asm block (228 k instructions per wavefront against the 32-bit kernel's 257 k, no v_addc), for everything up to 2^18.  Reads the scratch bign_prep_kernel<8> wrote (affine 1Q..8Q in
// 32-bit words, converted on load: 2 instructions per limb) and leaves (X, Z) for bign_inv_kernel like the big
// kernel.  Exceptional cases: every one of them (T = O, T = +-E, table point O) zeroes Z3 = Z1 H and all Z after it,
// so ONE test of the final Z sends the same signatures to bign_slow_kernel as the per-addition flags of
// bign_main_kernel do.
__global__ __launch_bounds__(256)
void bign_main29_kernel(size_t n, VerifyScratch S, const uint4 *__restrict__ gtab)
{
    constexpr int N = 8, NW = N / 2 + 1, W = Comb<N>::W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (S.status[idx] != ST_PENDING) return;
    uint32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = S.w[(size_t)i * S.n_pad + idx];

    jac29 T;
    affT<N> E;
    aff29 E29;
    load_qaff(E, S, (int)(w[NW - 1] & 15u) - 9, idx);       // top digit: 1 or 2
    f29_from_words(T.X, E.x);
    f29_from_words(T.Y, E.y);
    f29_set_one(T.Z);
    feT<N> u;
    load_soa(u, S.u, S.n_pad, idx);

    // 4N digits of v (4 doublings + one addition from the table of Q each), then the 2N comb windows of u, through
    // ONE addition site (the code of this kernel is what a lone wavefront pays for).
    // (round 5) A lone wavefront has nobody to hide a load behind: the table point of a digit is asked for BEFORE its four
    // doublings, the comb point of a window during the addition of the step before it (En) -- 49 trips to memory (the comb table is
    // 64 MiB: HBM / Infinity-Cache latency) used to sit directly in front of the addition that needs them.
    affT<N> En;
    bool have_n = false;
#pragma unroll 1
    for (int it = 4 * N - 1 + 32 * N / W; it >= 0; --it) {
        bool have;
        bool negate = false;
        if (it >= 32 * N / W) {
            const int d = (int)(w[NW - 2] >> 28) - 8;
#pragma unroll
            for (int l = NW - 2; l > 0; --l) w[l] = (w[l] << 4) | (w[l - 1] >> 28);
            w[0] <<= 4;
            have = d != 0;
            negate = d < 0;
            if (have) load_qaff(E, S, (d < 0 ? -d : d) - 1, idx);
#pragma unroll 1
            for (int k = 0; k < 4; ++k) jac29_dbl(T);
        } else {
            have = have_n;
            E = En;
        }
        if (it >= 1 && it - 1 < 32 * N / W) {                  // the next step is a comb window: its point, behind this step's addition
            const int win = 32 * N / W - 1 - (it - 1);
            const uint32_t b = u.v[0] & ((1u << W) - 1u);
#pragma unroll
            for (int l = 0; l < N - 1; ++l) u.v[l] = (u.v[l] >> W) | (u.v[l + 1] << (32 - W));
            u.v[N - 1] >>= W;
            have_n = b != 0;
            if (have_n) load_aff(En, gtab + ((size_t)win * (1u << W) + b) * (N / 2));
        }
        if (have) {
            f29_from_words(E29.x, E.x);
            f29_from_words(E29.y, E.y);
            if (negate) f29_neg(E29.y, E29.y);
            jac29_madd(T, E29);
        }
    }
    feT<N> X, Z;
    f29_to_words(Z, T.Z);
    if (fe_is_zero(Z)) { S.status[idx] = ST_SLOW; return; }
    f29_to_words(X, T.X);
    store_soa(S.rx, S.n_pad, idx, X);
    store_soa(S.u, S.n_pad, idx, Z);
}

// ------------------------------------------------- prep + main, smallest batches ---
// One signature per DPP quad (bign_quad29.hpp): up to 2^15 signatures the lanes are there for the taking (2^15 x 4
// lanes = two wavefronts per SIMD), and four lanes walk the 128 dependent doublings 2.6x as fast as one
// (tools/ubench/quad_dbl.hip).  The kernel does what prep + main do for the larger batches: range checks and
// scalars (every lane of the quad, the first one writes), the table 1Q..8Q -- kept JACOBIAN with Z^2 beside it: with
// spare lanes the general addition costs the same four levels as the mixed one, so no normalisation and no
// inversion -- in LDS as [entry][X, Y, Z, ZZ][limb][signature], each field element written by one lane of the quad;
// then 4N+1 signed radix-16 digits of v and the 2N comb windows of u through ONE addition site.  Leaves (X, Z) for
// bign_inv_kernel.  Exceptional cases zero Z (bign_quad29.hpp) and go to bign_slow_kernel.
// LANES = 4: a quad per signature (n <= 2^14); LANES = 2: a pair (2^14 < n <= 2^15, where quads would already put two
// wavefronts on a SIMD: 718 us against the 370 us of a lone wavefront; pairs keep one wavefront per SIMD up to 2^15).
// WG = 64 (one wavefront) up to 2^13 signatures; 256 above, so that the four wavefronts of a workgroup land on the
// four SIMDs of a CU.  LDS: 1152 B per signature.
template <int N, int WG, int LANES>
__global__ __launch_bounds__(WG)
void bign_quad29_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                        const uint8_t *__restrict__ pubkeys, size_t n, VerifyScratch S, const uint4 *__restrict__ gtab)
{
    // LANES = 8: a quad plus a HELPER quad per signature (up to 2^13 signatures the lanes are there): the helper adds up
    // the comb windows of u while the main quad walks v Q -- 16 of the 49 additions leave the critical path -- and hands
    // its point over at the end (one more addition).  Both quads run the same instructions: the helper's additions
    // ride on the main quad's, its doublings are masked off.
    constexpr int NW = N / 2 + 1, W = Comb<N>::W, NS = WG / LANES, L = LZ<N>::L, QL = LANES == 8 ? 4 : LANES;
    constexpr bool HELPER = LANES == 8;
    extern __shared__ int32_t s_tab[];              // [8 entries][X, Y, Z, ZZ][L limbs][NS signatures]
    const uint32_t q = threadIdx.x % QL, sl = threadIdx.x / LANES;
    const bool helper = HELPER && ((threadIdx.x >> 2) & 1u);
    const size_t idx = (size_t)blockIdx.x * NS + sl;
    if (idx >= n) return;                           // whole quads leave together
    affT<N> Q;
    feT<N> u;
    uint32_t w[NW];
    if (!prep_scalars<N>(hashes, sigs, pubkeys, idx, S, q == 0 && !helper, Q, u, w)) return;

    // the lanes of a signature share the writes: lane q owns field element q (X, Y, Z, ZZ) of every table entry
    // (a pair: q and q + 2)
    const auto put = [&](int e, const lqjacT<N> &P) {
        if (helper) return;
#pragma unroll
        for (int k = 0; k < 4 / QL; ++k) {
            const int fi = (int)q + QL * k;
            // (selected limb by limb, by VALUE: a reference picked by the lane's number made the compiler park the whole point in
            // scratch and index it there -- 436 / 676 / 916 bytes of scratch in the quad kernels, a round trip to memory per entry)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int32_t v = fi == 0 ? P.X.l[l] : fi == 1 ? P.Y.l[l] : fi == 2 ? P.Z.l[l] : P.D.l[l];
                s_tab[((e * 4 + fi) * L + l) * NS + sl] = v;
            }
        }
    };
    const auto dbl = [&](lqjacT<N> &P) { if constexpr (QL == 4) quad29_dbl(P, q); else pair29_dbl(P, q); };
    const auto add = [&](lqjacT<N> &P, const lqentT<N> &Q2) { if constexpr (QL == 4) quad29_add(P, Q2, q); else pair29_add(P, Q2, q); };
    const auto get = [&](lqentT<N> &E, int e) {
        const int32_t *b = s_tab + (size_t)e * 4 * L * NS + sl;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            E.X.l[l] = b[(0 * L + l) * NS]; E.Y.l[l] = b[(1 * L + l) * NS];
            E.Z.l[l] = b[(2 * L + l) * NS]; E.ZZ.l[l] = b[(3 * L + l) * NS];
        }
    };

    lqentT<N> E;                                    // Q as a table entry: affine
    f29_from_words(E.X, Q.x);
    f29_from_words(E.Y, Q.y);
    f29_set_one(E.Z);
    f29_set_one(E.ZZ);
    lqjacT<N> A, T, Wk;
    A.X = E.X; A.Y = E.Y; A.Z = E.Z; A.D = E.Z;
    T = A;
    put(0, A);
    // 2Q..8Q as two chains (A: 2Q 4Q 8Q, T: 3Q 6Q 7Q, 5Q = 4Q + Q), one doubling site and one addition site:
    // step = (operation, source chain, destination chain, entry)
#pragma unroll 1
    for (int step = 0; step < 7; ++step) {
        // step:      0      1      2      3      4      5      6
        // op:       dbl    add    dbl    dbl    add    add    dbl
        // source:    A      A      A      T      T      A      A
        // dest:      A      T      A      T      T      T      A
        // entry:     1      2      3      5      6      4      7
        const bool is_add = (0x32 >> step) & 1, from_t = (0x18 >> step) & 1, to_t = (0x3A >> step) & 1;
        const int entry = (int)((0x7465321u >> (4 * step)) & 15u);
        Wk = from_t ? T : A;
        if (is_add) add(Wk, E);
        else dbl(Wk);
        put(entry, Wk);
        if (to_t) T = Wk; else A = Wk;
    }
    // The table entries of a signature are written and read by the lanes of ONE wavefront (LANES <= 8 consecutive lanes), and
    // a wavefront's DS operations execute in order: all that is needed is that the writes have been issued before the reads
    // -- no workgroup barrier, which wavefronts that left early (idx >= n, a refused signature) would not have reached
    // (formally undefined in HIP, even though gfx9 drops terminated wavefronts from the barrier count).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // top digit d_{4N} = nibble_{4N}(w) - 8 is 1 or 2
    get(E, (int)(w[NW - 1] & 15u) - 9);
    T.X = E.X; T.Y = E.Y; T.Z = E.Z; T.D = E.ZZ;
    bool empty = helper;                            // the helper's sum starts at O: its first point is copied, not added
    affT<N> Gn;                                     // (no helper quad) the comb point asked for ahead, and its window value
    uint32_t bn = 0;
#pragma unroll 1
    for (int it = 4 * N - 1 + (HELPER ? 0 : 32 * N / W); it >= 0; --it) {
        const bool digit_step = HELPER || it >= 32 * N / W;
        const int win = HELPER ? 4 * N - 1 - it : 32 * N / W - 1 - it;      // the comb window of this step (helper: one per digit step)
        // a comb window of u: affine point from the table of G.  The helper asks for it FIRST, so that the main quad's
        // doublings cover the trip to memory.
        affT<N> G;
        uint32_t b = 0;
        if constexpr (HELPER) {
            if (helper && win < 32 * N / W) {
                b = u.v[0] & ((1u << W) - 1u);
#pragma unroll
                for (int l = 0; l < N - 1; ++l) u.v[l] = (u.v[l] >> W) | (u.v[l + 1] << (32 - W));
                u.v[N - 1] >>= W;
                if (b != 0) load_aff(G, gtab + ((size_t)win * (1u << W) + b) * (N / 2));
            }
        } else {
            // (round 5) no helper quad: the comb point of a window is asked for one step AHEAD -- during the addition of the step
            // before it -- instead of directly in front of its own addition (2N trips to a 64 MiB table on a lone wavefront)
            constexpr bool AHEAD = LANES == 2 || N != 8;       // measured: pairs -5.5 %, wide quads -2..-3 %, 256-bit quads +1 % (worse)
            if (!AHEAD) {
                if (!digit_step) {
                    b = u.v[0] & ((1u << W) - 1u);
#pragma unroll
                    for (int l = 0; l < N - 1; ++l) u.v[l] = (u.v[l] >> W) | (u.v[l + 1] << (32 - W));
                    u.v[N - 1] >>= W;
                    if (b != 0) load_aff(G, gtab + ((size_t)win * (1u << W) + b) * (N / 2));
                }
            } else if (!digit_step) { G = Gn; b = bn; }
            if (AHEAD && it >= 1 && it - 1 < 32 * N / W) {
                const int wn = 32 * N / W - 1 - (it - 1);
                bn = u.v[0] & ((1u << W) - 1u);
#pragma unroll
                for (int l = 0; l < N - 1; ++l) u.v[l] = (u.v[l] >> W) | (u.v[l + 1] << (32 - W));
                u.v[N - 1] >>= W;
                if (bn != 0) load_aff(Gn, gtab + ((size_t)wn * (1u << W) + bn) * (N / 2));
            }
        }
        bool have = b != 0;
        if (digit_step && !helper) {                // a digit of v: 4 doublings, then +- |d| Q from the LDS table
#pragma unroll 1
            for (int k = 0; k < 4; ++k) dbl(T);
            const int d = (int)(w[NW - 2] >> 28) - 8;
#pragma unroll
            for (int l = NW - 2; l > 0; --l) w[l] = (w[l] << 4) | (w[l - 1] >> 28);
            w[0] <<= 4;
            have = d != 0;
            if (have) {
                get(E, (d < 0 ? -d : d) - 1);
                if (d < 0) f29_neg(E.Y, E.Y);
            }
        } else if (have) {
            f29_from_words(E.X, G.x);
            f29_from_words(E.Y, G.y);
            f29_set_one(E.Z);
            f29_set_one(E.ZZ);
        }
        if (have) {
            if (empty) { T.X = E.X; T.Y = E.Y; T.Z = E.Z; T.D = E.ZZ; empty = false; }
            else add(T, E);
        }
    }
    if constexpr (HELPER) {                         // main quad: T += the helper's sum (lanes + 4)
        q29_from_next_quad(E.X, T.X);
        q29_from_next_quad(E.Y, T.Y);
        q29_from_next_quad(E.Z, T.Z);
        q29_from_next_quad(E.ZZ, T.D);
        const int helper_empty = __shfl_down((int)empty, 4);
        if (helper) return;
        if (!helper_empty) add(T, E);
    }
    feT<N> X, Z;
    f29_to_words(Z, T.Z);
    const bool zero = fe_is_zero(Z);
    if (q == 0) {
        if (zero) {
            S.status[idx] = ST_SLOW;
        } else {
            f29_to_words(X, T.X);
            store_soa(S.rx, S.n_pad, idx, X);
            store_soa(S.u, S.n_pad, idx, Z);
            S.status[idx] = ST_PENDING;
        }
    }
}

// --------------------------------------------------------------------- slow ---
// complete, branchy double-and-add for flagged lanes: R = u G + v Q, every exceptional
// case of ecpAddJ / ecpDblJA3 handled.  Rare (never for honest inputs).
__device__ __forceinline__ bool bit_at(const uint32_t *k, int i) { return (k[i >> 5] >> (i & 31)) & 1u; }

template <int N>
__global__ __launch_bounds__(64)
void bign_slow_kernel(const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ pubkeys,
                      size_t n, VerifyScratch S, size_t pk_stride)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (S.status[idx] != ST_SLOW) return;
    constexpr int NO = 4 * N;

    uint32_t u[N], v[N / 2 + 1];
    for (int i = 0; i < N; ++i) u[i] = S.u[(size_t)i * S.n_pad + idx];
    const uint8_t *sig = sigs + (NO + NO / 2) * idx;
    for (int i = 0; i < N / 2; ++i) v[i] = *reinterpret_cast<const uint32_t *>(sig + 4 * i);
    v[N / 2] = 1u;

    jacT<N> G, Q, T;
    fe_set_zero(G.X);
    for (int i = 0; i < N; ++i) G.Y.v[i] = curve_yG<N>()[i];
    fe_set_one(G.Z);
    if (pk_stride == ~(size_t)0) {                             // K signers: the main kernel left Q in qtab row 0
        affT<N> A;
        load_qaff(A, S, 0, idx);
        Q.X = A.x; Q.Y = A.y;
    } else {
        load_fe_bytes(Q.X, pubkeys + pk_stride * idx);         // pk_stride = 2 NO, or 0 for a batch under one key
        load_fe_bytes(Q.Y, pubkeys + pk_stride * idx + NO);
    }
    fe_set_one(Q.Z);
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
#pragma unroll 1
    for (int i = 32 * N - 1; i >= 0; --i) {
        jac_dbl(T);
        if (bit_at(u, i)) jac_add_complete(T, G);
        if (i <= 16 * N && bit_at(v, i)) jac_add_complete(T, Q);
    }
    if (fe_is_zero(T.Z)) { S.status[idx] = ERR_BAD_SIG; return; }      // R == O (bign_sign.c:332-336)
    feT<N> zi = fe_inv(T.Z);
    fe_sqr(zi, zi);
    fe_mul(zi, T.X, zi);
    fe_canon(zi, zi);
    store_soa(S.rx, S.n_pad, idx, zi);
    fe_set_one(zi);
    store_soa(S.u, S.n_pad, idx, zi);                  // already affine: Z = 1 for bign_inv_kernel
    S.status[idx] = ST_PENDING;
}

// ---------------------------------------------------------------- inversion ---
// x_R = X / Z^2 for all pending signatures with Montgomery's simultaneous inversion: lane j owns the K
// signatures j, j + lanes, j + 2 lanes, ... (coalesced in the limb-major scratch), multiplies their Z
// (prefix products parked in the dead qz rows), inverts the product once (255 S + 13 M on the
// 256-bit curve, as much as 1/7 of a whole verification when done per signature) and peels the
// individual inverses off with two multiplications each.  Z != 0 for every pending lane (main / slow
// send R = O elsewhere), so the product is invertible.  Same x_R as ecpToAJ: the canonical residue.
template <int N>
__global__ __launch_bounds__(64)
void bign_inv_kernel(size_t n, size_t lanes, int K, VerifyScratch S)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lanes) return;
    if constexpr (N != 8) {
    // (the wider curves: loads in front of their use, as in rounds 1-4 -- asking ahead did not pay there, profiles/r05_ask_ahead_ab.txt)
    uint32_t *Z = S.u, *PZ = S.qz;
    feT<N> acc, z;
    fe_set_one(acc);
#pragma unroll 1
    for (int t = 0; t < K; ++t) {
        const size_t idx = (size_t)t * lanes + j;
        if (idx >= n) break;
        if (S.status[idx] == ST_PENDING) {
            load_soa(z, Z, S.n_pad, idx);
            fe_mul<1, VtOps>(acc, acc, z);
        }
        store_soa(PZ, S.n_pad, idx, acc);             // product of the pending Z up to and including t
    }
    feT<N> inv = fe_inv_checked(acc);
#pragma unroll 1
    for (int t = K - 1; t >= 0; --t) {
        const size_t idx = (size_t)t * lanes + j;
        if (idx >= n || S.status[idx] != ST_PENDING) continue;
        feT<N> zi = inv;
        if (t > 0) {
            load_soa(acc, PZ, S.n_pad, idx - lanes);
            fe_mul<1, VtOps>(zi, inv, acc);                     // 1 / Z_t
            load_soa(z, Z, S.n_pad, idx);
            fe_mul<1, VtOps>(inv, inv, z);                      // 1 / (product up to t - 1)
        }
        fe_sqr<1, VtOps>(zi, zi);
        load_soa(z, S.rx, S.n_pad, idx);
        fe_mul<1, VtOps>(zi, z, zi);
        fe_canon(zi, zi);
        store_soa(S.rx, S.n_pad, idx, zi);
    }
    } else {
    uint32_t *Z = S.u, *PZ = S.qz;
    // (round 5: every load is asked for one step ahead of the multiplication that needs it -- this kernel is a few lone wavefronts,
    //  valu_busy 0.30, and its loads used to sit directly in front of their use)
    feT<N> acc, z, zn, pn, xn;
    fe_set_one(acc);
    const auto pending = [&](int t) { const size_t i = (size_t)t * lanes + j; return t >= 0 && t < K && i < n && S.status[i] == ST_PENDING; };
    if (pending(0)) load_soa(zn, Z, S.n_pad, j);
#pragma unroll 1
    for (int t = 0; t < K; ++t) {
        const size_t idx = (size_t)t * lanes + j;
        if (idx >= n) break;
        const bool mine = S.status[idx] == ST_PENDING;
        z = zn;
        if (pending(t + 1)) load_soa(zn, Z, S.n_pad, idx + lanes);
        if (mine) fe_mul<1, VtOps>(acc, acc, z);
        store_soa(PZ, S.n_pad, idx, acc);             // product of the pending Z up to and including t
    }
    feT<N> inv = fe_inv_checked(acc);
    const auto ask = [&](int t) {                      // operands of step t: the product up to t - 1, Z_t, X_t
        const size_t i = (size_t)t * lanes + j;
        if (t > 0) { load_soa(pn, PZ, S.n_pad, i - lanes); load_soa(zn, Z, S.n_pad, i); }
        load_soa(xn, S.rx, S.n_pad, i);
    };
    int t = K - 1;
    while (t >= 0 && !pending(t)) --t;
    if (t >= 0) ask(t);
#pragma unroll 1
    while (t >= 0) {
        const size_t idx = (size_t)t * lanes + j;
        const feT<N> p = pn, x = xn;
        z = zn;
        int nt = t - 1;
        while (nt >= 0 && !pending(nt)) --nt;
        if (nt >= 0) ask(nt);
        feT<N> zi = inv;
        if (t > 0) {
            fe_mul<1, VtOps>(zi, inv, p);                       // 1 / Z_t
            fe_mul<1, VtOps>(inv, inv, z);                      // 1 / (product up to t - 1)
        }
        fe_sqr<1, VtOps>(zi, zi);
        fe_mul<1, VtOps>(zi, x, zi);
        fe_canon(zi, zi);
        store_soa(S.rx, S.n_pad, idx, zi);
        t = nt;
    }
    }
}

// --------------------------------------------------------------------- tail ---
// The DER OID that leads the hashed message travels as a kernel argument.  bee2 accepts any valid DER OID (bign_sign.c:
// 268-300), so a longer one is split: its whole 32-byte blocks -- the same for every signature of the batch -- are absorbed
// ONCE by the streaming belt-hash kernel into a 48-byte state h || s in device scratch (hs0, pre_len octets), and the
// per-signature kernels start from that state with the remaining len < 32 octets in der[].
constexpr int OID_MAX = 128;                      // longest OID piece a kernel stages itself
struct OidArg {
    uint32_t len;                 // octets in der[]
    uint32_t pre_len;             // octets already absorbed into *hs0 (a multiple of 32; 0 = none)
    const uint32_t *hs0;          // device: belt-hash state h[8] || s[4] after pre_len octets (valid when pre_len != 0)
    uint8_t der[OID_MAX];
};
// belt-hash state of a lane at the start of its message: the standard's initial value (belt_hash.c:52) or the OID prefix's
__device__ __forceinline__ void oid_hash_start(uint32_t (&h)[8], uint32_t (&s)[4], const OidArg &oid)
{
    if (oid.pre_len) {
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = oid.hs0[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = oid.hs0[8 + i];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            h[i] = (uint32_t)c_beltH[4 * i] | (uint32_t)c_beltH[4 * i + 1] << 8 |
                   (uint32_t)c_beltH[4 * i + 2] << 16 | (uint32_t)c_beltH[4 * i + 3] << 24;
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = 0;
    }
}
err_t make_oid_arg(OidArg &oa, const uint8_t *oid_der, size_t oid_len, hipStream_t st)
{
    memset(&oa, 0, sizeof oa);
    size_t pre = 0;
    if (oid_len > OID_MAX) {
        if (oid_len >> 28) return ERR_BAD_OID;            // the bit length is carried in 32 bits below
        pre = oid_len / 32 * 32;
        void *scr = nullptr;
        err_t code = scratch_for_stream(st, 12, 48 + pre, &scr);
        if (code != ERR_OK) return code;
        uint32_t init[12];
        const uint8_t *H = host_beltH();
        for (int i = 0; i < 8; ++i)
            init[i] = (uint32_t)H[4 * i] | (uint32_t)H[4 * i + 1] << 8 | (uint32_t)H[4 * i + 2] << 16 | (uint32_t)H[4 * i + 3] << 24;
        for (int i = 8; i < 12; ++i) init[i] = 0;
        // pageable sources: the runtime has taken its copy when these calls return
        B2H_TRY(hipMemcpyAsync(scr, init, 48, hipMemcpyHostToDevice, st));
        B2H_TRY(hipMemcpyAsync((uint8_t *)scr + 48, oid_der, pre, hipMemcpyHostToDevice, st));
        code = launch_belt_hash_stream(scr, (uint8_t *)scr + 48, pre / 32, 0, 0, 0, st);
        if (code != ERR_OK) return code;
        oa.hs0 = (const uint32_t *)scr;
        oa.pre_len = (uint32_t)pre;
    }
    oa.len = (uint32_t)(oid_len - pre);
    memcpy(oa.der, oid_der + pre, oa.len);
    return ERR_OK;
}

// Tab = BeltTabSmall: 64-thread workgroups, 4 KiB table with bank conflicts (any curve, any batch size);
// Tab = BeltTabTwo: 1024-thread workgroups around the conflict-free 64 KiB table of the CTR kernel -- worth its
// fill for big batches on the 256-bit curve (the rows of the wider curves do not fit beside the table).
// Dynamic LDS: table first (BeltTabTwo composes addresses with OR and needs its 64 KiB alignment), rows behind.
template <int N, class Tab, int WG>
__global__ __launch_bounds__(WG)
void bign_tail_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                      size_t n, VerifyScratch S, OidArg oid, uint32_t *__restrict__ codes)
{
    constexpr int NO = 4 * N;
    constexpr int DW = 2 * N;                              // words of <x_R> || H
    constexpr int LSTR = DW + 1;                           // odd word stride: conflict-free per-lane rows
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint8_t *s_tab = s_dyn;
    uint32_t *s_dat = reinterpret_cast<uint32_t *>(s_dyn + Tab::kBytes);
    Tab::fill(s_tab, threadIdx.x, WG);
    __syncthreads();
    const Tab T(s_tab);

    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t st = S.status[idx];
    if (st != ST_PENDING) { codes[idx] = st; return; }

    // message = oid_der || <x_R> || H  (bign_sign.c:339-342).  <x_R> || H sits in LDS as whole words, one
    // row per lane; the OID is wavefront-uniform, so word k of the message is either an OID word or two
    // neighbouring data words funnel-shifted by the OID length mod 4 -- no byte traffic.
    uint32_t *d = s_dat + threadIdx.x * LSTR;
#pragma unroll
    for (int l = 0; l < N; ++l) d[l] = S.rx[(size_t)l * S.n_pad + idx];
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
        const uint4 a = *reinterpret_cast<const uint4 *>(hashes + NO * idx + 16 * k);
        d[N + 4 * k] = a.x; d[N + 4 * k + 1] = a.y; d[N + 4 * k + 2] = a.z; d[N + 4 * k + 3] = a.w;
    }
    const uint32_t L = oid.len + 2 * NO;
    const uint32_t q = oid.len >> 2, r8 = (oid.len & 3u) * 8u;
    auto oid_word = [&](uint32_t k) -> uint32_t {          // bytes 4k .. 4k+3 of the OID, zero past its end
        uint32_t w = 0;
        for (uint32_t b = 0; b < 4; ++b)
            if (4 * k + b < oid.len) w |= (uint32_t)oid.der[4 * k + b] << (8 * b);
        return w;
    };
    const uint32_t oid_tail = r8 ? oid_word(q) << (32 - r8) : 0u;   // the OID's last 1..3 bytes, top-aligned
    auto msg_word = [&](uint32_t k) -> uint32_t {
        if (k < q) return oid_word(k);
        const uint32_t t = k - q;
        const uint32_t hi = t < (uint32_t)DW ? d[t] : 0u;
        const uint32_t lo = t == 0 ? oid_tail : (t - 1 < (uint32_t)DW ? d[t - 1] : 0u);
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> (32 - r8));
    };

    // belt-hash (src/crypto/belt/belt_hash.c:43-171)
    uint32_t h[8], s[4], X[8], s1[4];
    oid_hash_start(h, s, oid);
    const uint32_t nblk = (L + 31) / 32;
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; ++b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) X[i] = msg_word(8 * b + i);
        belt_compress(T, s1, h, X);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] ^= s1[i];
    }
    // final block: <bit length>_128 || s  (belt_hash.c:120-135, belt_lcl.c:25-51)
    X[0] = (L + oid.pre_len) << 3; X[1] = (L + oid.pre_len) >> 29; X[2] = 0; X[3] = 0;
    X[4] = s[0]; X[5] = s[1]; X[6] = s[2]; X[7] = s[3];
    belt_compress(T, s1, h, X);
    // the first l bits = N/2 words of the hash must equal s0 (beltHashStepV2(sig, no/2, ..))
    const uint32_t *s0 = reinterpret_cast<const uint32_t *>(sigs + (NO + NO / 2) * idx);
    bool match = true;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) match = match && (h[i] == s0[i]);
    codes[idx] = match ? ERR_OK : ERR_BAD_SIG;
}

// ------------------------------------------------------------------- G table ---
// 8-bit table: entry (win, b) = b * 2^(8 win) * G in affine form, b = 1..255.  One thread per
// entry, complete double-and-add; runs once per device.
template <int N>
__global__ __launch_bounds__(64)
void bign_gtable_kernel(uint4 *__restrict__ gtab)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 4 * N * GT8_ENTRIES) return;
    const int win = id / GT8_ENTRIES, b = id % GT8_ENTRIES;
    uint4 *e = gtab + (size_t)id * (N / 2);
    if (b == 0) { for (int k = 0; k < N / 2; ++k) e[k] = make_uint4(0, 0, 0, 0); return; }
    jacT<N> G, T;
    fe_set_zero(G.X);
    for (int i = 0; i < N; ++i) G.Y.v[i] = curve_yG<N>()[i];
    fe_set_one(G.Z);
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
    const int top = 8 * win + 7;
#pragma unroll 1
    for (int i = top; i >= 0; --i) {
        jac_dbl(T);
        const int rel = i - 8 * win;
        if (rel >= 0 && ((b >> rel) & 1)) jac_add_complete(T, G);
    }
    feT<N> x, y;
    to_affine(x, y, T);
    store_aff(e, x, y);
}

// 16-bit table from the 8-bit one: entry16(w, b) = entry8(2w, b & 255) + entry8(2w+1, b >> 8).
template <int N>
__global__ __launch_bounds__(256)
void bign_gtable16_kernel(const uint4 *__restrict__ gtab8, uint4 *__restrict__ gtab16)
{
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)(2 * N) * 65536) return;
    const int win = (int)(id >> 16), b = (int)(id & 65535);
    const int lo = b & 255, hi = b >> 8;
    uint4 *e = gtab16 + id * (N / 2);
    if (b == 0) { for (int k = 0; k < N / 2; ++k) e[k] = make_uint4(0, 0, 0, 0); return; }
    const uint4 *elo = gtab8 + ((size_t)(2 * win) * GT8_ENTRIES + lo) * (N / 2);
    const uint4 *ehi = gtab8 + ((size_t)(2 * win + 1) * GT8_ENTRIES + hi) * (N / 2);
    if (hi == 0 || lo == 0) {                     // one summand is the neutral element: copy
        const uint4 *src = hi == 0 ? elo : ehi;
        for (int k = 0; k < N / 2; ++k) e[k] = src[k];
        return;
    }
    affT<N> A, B;
    load_aff(A, elo);
    load_aff(B, ehi);
    jacT<N> T, E;
    T.X = A.x; T.Y = A.y; fe_set_one(T.Z);
    E.X = B.x; E.Y = B.y; fe_set_one(E.Z);
    jac_add_complete(T, E);                       // distinct multiples of G below the group order: never O
    feT<N> x, y;
    to_affine(x, y, T);
    store_aff(e, x, y);
}

// ------------------------------------------------------ one signer (round 4) ---
// Many signatures under ONE public key -- the shape of `bee2cmd sig vfy` over a tree of files (cmd/core/cmd_sig.c:484-490), of a
// signed log or package repository.  Q is then a FIXED base like G: a comb table of Q (8-bit windows, entry (win, b) =
// b 2^(8 win) Q affine, win = 0 .. 2N: v = s0 + 2^l has 16 N + 1 bits) replaces the 16 N doublings, the 4 N additions on the
// signed radix-16 digits and the per-signature table 1Q..8Q with its normalisation of bign_prep / bign_main: R = u G + v Q
// is 2 N mixed additions for v (the top window is always 1: the accumulator starts there) and 2 N for u on the 16-bit
// table of G -- 32 instead of 128 doublings + 48 additions on the 256-bit curve.  The table (278 / 612 / 1 072 KiB) is built
// once per (device, key) from the 2N + 1 points 2^(8 win) Q the host hands over (a chain of 16 N doublings: 25 us on a host
// core, 0.5 ms for a lone wavefront) and cached.  Same verdicts as bign_main: exceptional additions send the signature to
// bign_slow_kernel (which reads the one key with stride 0).
// blockIdx.y = the key of a slab of tables being built together (table y at ktab + y * tab_quads, its starting points at base + y *
// base_quads): the keys a call meets for the first time share one allocation, one upload and this one launch.
template <int N>
__global__ __launch_bounds__(64)
void bign_ktable_kernel(const uint4 *__restrict__ base, uint4 *__restrict__ ktab, size_t base_quads, size_t tab_quads)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (2 * N + 1) * GT8_ENTRIES) return;
    base += (size_t)blockIdx.y * base_quads;
    ktab += (size_t)blockIdx.y * tab_quads;
    const int win = id / GT8_ENTRIES, b = id % GT8_ENTRIES;
    uint4 *e = ktab + (size_t)id * (N / 2);
    if (b == 0) { for (int k = 0; k < N / 2; ++k) e[k] = make_uint4(0, 0, 0, 0); return; }
    affT<N> B;
    load_aff(B, base + (size_t)win * (N / 2));
    jacT<N> P, T;
    P.X = B.x; P.Y = B.y; fe_set_one(P.Z);
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
#pragma unroll 1
    for (int i = 7; i >= 0; --i) {
        jac_dbl(T);
        if ((b >> i) & 1) jac_add_complete(T, P);
    }
    // b 2^(8 win) < q and Q is a point of the (prime-order) curve -- the host checked -- so T != O
    feT<N> x, y;
    to_affine(x, y, T);
    store_aff(e, x, y);
}

// one lane per signature: range checks and scalars (prep_scalars with the one key), then the two combs; leaves (X, Z) of R for
// bign_inv_kernel like bign_main_kernel.  key = the public key in device memory (2 NO octets, behind the table).
// Q16: the windows of v are 16 bits wide and read from ktab16 (entry (w, b) = b 2^(16 w) Q, w = 0 .. N - 1: the table a key gets
// once enough signatures have been verified under it, bign_key_table) -- N instead of 2N additions for v.
// KEYED: K signers -- key = K public keys (2 NO octets each), key_index[i] < nkeys says whose signature i is, tabs[k] = the 8-bit
// table of key k, or null for a key that is not a point of the curve: such signatures go to bign_slow_kernel, which like bee2 does not
// look (it finds Q in qtab row 0, where the general path keeps it).  An index out of range is ERR_BAD_INPUT for that signature.
template <int N, class OPS = VtOps, bool Q16 = false, bool KEYED = false>
__global__ __launch_bounds__(256, (N == 8 ? 4 : N == 12 ? 1 : 2))
void bign_onekey_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ key,
                        size_t n, VerifyScratch S, const uint4 *__restrict__ gtab, const uint4 *__restrict__ ktab,
                        const uint4 *__restrict__ ktab16, const uint32_t *__restrict__ key_index,
                        const uint4 *const *__restrict__ tabs, uint32_t nkeys)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    constexpr int W = Comb<N>::W;
    constexpr int NO = 4 * N;
    if (KEYED) {
        const uint32_t k = key_index[idx];
        if (k >= nkeys) { S.status[idx] = ERR_BAD_INPUT; return; }
        key += (size_t)2 * NO * k;
        ktab = tabs[k];
        ktab16 = tabs[nkeys + k];                                  // the 16-bit table of a busy key, or null
    }
    feT<N> u;
    {
        affT<N> Q;
        uint32_t w[N / 2 + 1];
        if (!prep_scalars<N>(hashes, sigs, key, idx, S, true, Q, u, w, 0)) return;      // status = the error code
        if (KEYED) {
            store_qxy(S, 0, idx, Q.x, Q.y);                                             // (for bign_slow_kernel)
            if (!ktab) { S.status[idx] = ST_SLOW; return; }
        }
    }
    bool ok = true;
    jacT<N> T;
    {
        affT<N> E;                                                           // the top window of v = s0 + 2^l is 1
        load_aff(E, ktab + ((size_t)(2 * N) * GT8_ENTRIES + 1) * (N / 2));
        T.X = E.x; T.Y = E.y; fe_set_one(T.Z);
    }
    const uint32_t *s0 = reinterpret_cast<const uint32_t *>(sigs + (NO + NO / 2) * idx);
    // KEYED: a lane whose key has a 16-bit table takes the word in two halves (steps 0 and 2, nothing at 1 and 3), the others in
    // four octets -- one loop for both, so a wavefront of busy keys skips the odd steps altogether
    const bool q16 = Q16 || (KEYED && ktab16 != nullptr);
#pragma unroll 1
    for (int l = 0; l < N / 2; ++l) {
        const uint32_t word = s0[l];
#pragma unroll 1
        for (int k = 0; k < ((Q16 && !KEYED) ? 2 : 4); ++k) {
            const uint4 *entry = nullptr;
            if (Q16 && !KEYED) {
                const uint32_t b = (word >> (16 * k)) & 65535u;
                if (b != 0) entry = ktab16 + ((size_t)(2 * l + k) * 65536 + b) * (N / 2);
            } else if (q16) {
                const uint32_t b = (word >> (8 * k)) & 65535u;
                if ((k & 1) == 0 && b != 0) entry = ktab16 + ((size_t)(2 * l + (k >> 1)) * 65536 + b) * (N / 2);
            } else {
                const uint32_t b = (word >> (8 * k)) & 255u;
                if (b != 0) entry = ktab + ((size_t)(4 * l + k) * GT8_ENTRIES + b) * (N / 2);
            }
            if (entry) {
                affT<N> E;
                load_aff(E, entry);
                ok &= jac_madd<N, OPS>(T, E);
            }
        }
    }
    // + u G : comb over the W-bit windows of u, as bign_main_kernel
#pragma unroll 1
    for (int win = 0; win < 32 * N / W; ++win) {
        const uint32_t b = u.v[0] & ((1u << W) - 1u);
#pragma unroll
        for (int l = 0; l < N - 1; ++l) u.v[l] = (u.v[l] >> W) | (u.v[l + 1] << (32 - W));
        u.v[N - 1] >>= W;
        if (b != 0) {
            affT<N> E;
            load_aff(E, gtab + ((size_t)win * (1u << W) + b) * (N / 2));
            ok &= jac_madd<N, OPS>(T, E);
        }
    }
    ok &= !fe_is_zero(T.Z);
    if (!ok) { S.status[idx] = ST_SLOW; return; }
    S.status[idx] = ST_PENDING;
    store_soa(S.rx, S.n_pad, idx, T.X);
    store_soa(S.u, S.n_pad, idx, T.Z);
}

// The same sum on FOUR lanes per signature, for batches that leave the chip's SIMDs underfed (<= 2^16 signatures): R = u G + v Q is a
// sum of 2N + N (+ 1) table points and nothing orders them -- lane `sub` of a quad adds the windows 4 i + sub of v, then of u (the top
// window of v, always 1, starts lane 0), and two rounds of Jacobian additions across the quad fold the four partial sums: 6 + 2 point
// additions deep instead of 24 on the 256-bit curve.  A lone wavefront pays for every instruction it issues, whatever the lane count
// (DESIGN 2), so this is a third of the main kernel's latency.  A lane's digits come from memory at lane-dependent addresses (the
// octets of s0; u from a private LDS row) -- a register array indexed by the lane would go to scratch.  Empty partial sums (all of a
// lane's windows zero) are flags, not points; exceptional additions send the signature to bign_slow_kernel as everywhere.
template <int N, class OPS = VtOps, bool Q16 = false, bool KEYED = false>
__global__ __launch_bounds__(256, (N == 8 ? 2 : 1))
void bign_onekey4_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ key,
                         size_t n, VerifyScratch S, const uint4 *__restrict__ gtab, const uint4 *__restrict__ ktab,
                         const uint4 *__restrict__ ktab16, const uint32_t *__restrict__ key_index,
                         const uint4 *const *__restrict__ tabs, uint32_t nkeys)
{
    __shared__ uint32_t u_rows[256][N + 1];                       // (+ 1: rows on different banks)
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned sub = threadIdx.x & 3u;
    size_t idx = gid >> 2;
    bool live = idx < n;                                           // dead quads walk the last signature and store nothing
    if (!live) idx = n - 1;
    constexpr int W = Comb<N>::W;
    constexpr int NO = 4 * N;
    if (KEYED) {
        const uint32_t k = key_index[idx];
        if (k >= nkeys) { if (live && sub == 0) S.status[idx] = ERR_BAD_INPUT; live = false; }
        else { key += (size_t)2 * NO * k; ktab = tabs[k]; ktab16 = tabs[nkeys + k]; }
    }
    {
        affT<N> Q;
        feT<N> u;
        uint32_t w[N / 2 + 1];
        const bool write = live && sub == 0;
        if (!prep_scalars<N>(hashes, sigs, key, idx, S, write, Q, u, w, 0)) live = false;          // status = the error code (lane 0)
        else if (KEYED) {
            if (write) store_qxy(S, 0, idx, Q.x, Q.y);                                              // (for bign_slow_kernel)
            if (!ktab) { if (write) S.status[idx] = ST_SLOW; live = false; }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) u_rows[threadIdx.x][i] = u.v[i];
    }
    if (KEYED && !ktab) { ktab = gtab; ktab16 = nullptr; }         // (a dead quad still walks: any readable table)
    bool ok = true, empty = sub != 0;
    jacT<N> T;
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
    if (sub == 0) {
        affT<N> E;                                                 // the top window of v = s0 + 2^l is 1
        load_aff(E, ktab + ((size_t)(2 * N) * GT8_ENTRIES + 1) * (N / 2));
        T.X = E.x; T.Y = E.y; fe_set_one(T.Z);
    }
    const auto take = [&](const uint4 *entry) {
        affT<N> E;
        load_aff(E, entry);
        if (empty) { T.X = E.x; T.Y = E.y; fe_set_one(T.Z); empty = false; }
        else ok &= jac_madd<N, OPS>(T, E);
    };
    const uint8_t *s0 = sigs + (NO + NO / 2) * idx;
    // windows of v below the top one: N of 16 bits (a key with a 16-bit table; KEYED: by the lane's key) or 2N of 8 bits
    const bool q16 = (Q16 && !KEYED) || (KEYED && ktab16 != nullptr);
    constexpr int NV_MAX = (Q16 && !KEYED) ? N : 2 * N;
    const int nv = q16 ? N : 2 * N;
#pragma unroll 1
    for (int i = 0; i < NV_MAX / 4; ++i) {
        const unsigned j = 4u * i + sub;
        if ((int)j >= nv) continue;
        if (q16) {
            const uint32_t b = reinterpret_cast<const uint16_t *>(s0)[j];
            if (b != 0) take(ktab16 + ((size_t)j * 65536 + b) * (N / 2));
        } else {
            const uint32_t b = s0[j];
            if (b != 0) take(ktab + ((size_t)j * GT8_ENTRIES + b) * (N / 2));
        }
    }
    static_assert(W == 16, "16-bit comb of G");
#pragma unroll 1
    for (int i = 0; i < 2 * N / 4; ++i) {
        const unsigned j = 4u * i + sub;
        const uint32_t b = (u_rows[threadIdx.x][j >> 1] >> (16u * (j & 1u))) & 65535u;
        if (b != 0) take(gtab + ((size_t)j * 65536 + b) * (N / 2));
    }
    // fold the quad: lanes sub ^ 1, then sub ^ 2
#pragma unroll 1
    for (int step = 1; step <= 2; step <<= 1) {
        jacT<N> P;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            P.X.v[l] = (uint32_t)__shfl_xor((int)T.X.v[l], step, 64);
            P.Y.v[l] = (uint32_t)__shfl_xor((int)T.Y.v[l], step, 64);
            P.Z.v[l] = (uint32_t)__shfl_xor((int)T.Z.v[l], step, 64);
        }
        const bool p_empty = __shfl_xor((int)empty, step, 64) != 0;
        ok &= __shfl_xor((int)ok, step, 64) != 0;
        if (!p_empty) {
            if (empty) { T = P; empty = false; }
            else ok &= jac_add<N, OPS>(T, P);
        }
    }
    if (!live || sub != 0) return;
    ok &= !empty && !fe_is_zero(T.Z);
    if (!ok) { S.status[idx] = ST_SLOW; return; }
    S.status[idx] = ST_PENDING;
    store_soa(S.rx, S.n_pad, idx, T.X);
    store_soa(S.u, S.n_pad, idx, T.Z);
}

// Table of the signing side's one-lane kernel (round 3): signed 6-bit windows, entry (i, j) = j 2^(6i) G, j = 1..32, at
// index i * 32 + j - 1; W6 = ceil((32N + 1) / 6) windows (43 / 65 / 86), the last one takes what is left of the scalar
// plus the carry of the recoding (at most 16 / 1 / 4).  From the seed table: j 2^(6i) = lo 2^(8a) + hi 2^(8a + 8); the one
// multiple outside it that a digit can ask for is 2^(32N) G = 2 (128 2^(32N - 8) G).  Entries no digit reaches are zero.
// (round 4) WB = 7: the same with signed 7-bit windows -- 64 entries per window, ceil((32N + 1) / 7) windows (37 on the 256-bit
// curve, whose top window starts at bit 252 like the 6-bit one's) -- for the kernel that looks its entry up in LDS
// (bign_mulbase_lds_kernel): the row size no longer costs a masked scan there, so fewer, wider windows pay.
template <int N> struct Win6 { static constexpr int W = (32 * N + 1 + 5) / 6; };
template <int N, int WB> struct WinW { static constexpr int W = (32 * N + 1 + WB - 1) / WB, ENT = 1 << (WB - 1); };
template <int N, int WB = 6>
__global__ __launch_bounds__(64)
void bign_gtable6_kernel(const uint4 *__restrict__ gtab8, uint4 *__restrict__ gtab6)
{
    constexpr int ENT = WinW<N, WB>::ENT;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= WinW<N, WB>::W * ENT) return;
    const int i = id / ENT, j = id % ENT + 1;
    const int a = (WB * i) >> 3, r = (WB * i) & 7;
    const int v = j << r, part[2] = {v & 255, v >> 8};
    uint4 *e = gtab6 + (size_t)id * (N / 2);
    jacT<N> T, E;
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
    bool reachable = true;
    for (int h = 0; h < 2; ++h) {
        const int m = part[h], win = a + h;
        if (m == 0) continue;
        affT<N> A;
        if (win < 4 * N) {
            load_aff(A, gtab8 + ((size_t)win * GT8_ENTRIES + m) * (N / 2));
            E.X = A.x; E.Y = A.y; fe_set_one(E.Z);
        } else if (win == 4 * N && m == 1) {
            load_aff(A, gtab8 + ((size_t)(4 * N - 1) * GT8_ENTRIES + 128) * (N / 2));
            E.X = A.x; E.Y = A.y; fe_set_one(E.Z);
            jac_dbl(E);
        } else {
            reachable = false;
            continue;
        }
        jac_add_complete(T, E);
    }
    if (!reachable) { for (int k = 0; k < N / 2; ++k) e[k] = make_uint4(0, 0, 0, 0); return; }
    feT<N> x, y;
    to_affine(x, y, T);
    fe_canon(x, x);
    fe_canon(y, y);
    store_aff(e, x, y);
}

// -------------------------------------------------------------- pubkey val ---
// bignPubkeyValEc (bign_misc.c:319-352): both coordinates < p (qrFrom) and the point on the curve,
// ecpIsOnA (src/math/ecp/ecp_a.c:36-60): (x^2 + a) x + b == y^2 with a = p - 3 on all three
// standard curves.  One lane per key; 8N octets in, one err_t out -- HBM-bound.
template <int N>
__global__ __launch_bounds__(256)
void bign_pubkey_val_kernel(const uint8_t *__restrict__ pubkeys, size_t n, uint32_t *__restrict__ codes)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    constexpr int NO = 4 * N;
    feT<N> x, y, t, b, three;
    load_fe_bytes(x, pubkeys + 2 * NO * idx);
    load_fe_bytes(y, pubkeys + 2 * NO * idx + NO);
    uint32_t P[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { P[i] = 0xFFFFFFFFu; b.v[i] = curve_b<N>()[i]; three.v[i] = 0; }
    P[0] = 0u - CurveC<N>::C;
    three.v[0] = 3u;
    bool ok = !limbs_ge(x.v, P) && !limbs_ge(y.v, P);
    fe_sqr(t, x);
    fe_sub(t, t, three);
    fe_mul(t, t, x);
    fe_add(t, t, b);
    fe_sqr(y, y);
    fe_sub(t, t, y);
    ok = ok && fe_is_zero(t);
    codes[idx] = ok ? ERR_OK : ERR_BAD_PUBKEY;
}

#ifdef BEE2HIP_EXPERIMENTS
// --------------------------------------------------------- debug / self-test ---
// element-wise field ops over arrays of N-limb values, used by tests/test_gpu_bign.py
// to check the GF(p) layer against Python big integers.  op: 0 mul, 1 sqr, 2 add, 3 sub,
// 4 inv, 5 mul<3>, 6 sqr<8>, 7 canon, 8 dbl-point-x (a = X, b = Y, Z = 1 -> affine x of 2P)
template <int N>
__global__ void bign_debug_fe_kernel(int op, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    feT<N> x, y, r;
    for (int i = 0; i < N; ++i) { x.v[i] = a[N * idx + i]; y.v[i] = b[N * idx + i]; }
    switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sqr(r, x); break;
    case 2: fe_add(r, x, y); break;
    case 3: fe_sub(r, x, y); break;
    case 4: r = fe_inv(x); break;
    case 11: fe_canon(x, x); r = fe_inv_safegcd(x); break;      // division steps alone, no fallback
    case 12: r = fe_inv_checked(x); break;
    case 5: fe_mul<3>(r, x, y); break;
    case 6: fe_sqr<8>(r, x); break;
    case 7: r = x; break;
    // 100 + op: the same operation in its verification flavour (VtOps: carry chains with rare wavefront-uniform branches)
    case 100: fe_mul<1, VtOps>(r, x, y); break;
    case 101: fe_sqr<1, VtOps>(r, x); break;
    case 102: fe_add<VtOps>(r, x, y); break;
    case 103: fe_sub<VtOps>(r, x, y); break;
    case 105: fe_mul<3, VtOps>(r, x, y); break;
    case 106: fe_sqr<8, VtOps>(r, x); break;
    case 104: fe_neg<VtOps>(r, x); break;
    case 109: case 110: {
        uint32_t w[2 * N];
        for (int i = 0; i < N; ++i) { w[i] = x.v[i]; w[N + i] = y.v[i]; }
        if (op == 109) fe_reduce<1, VtOps>(r, w); else fe_reduce<3, VtOps>(r, w);
    } break;
    case 108: {
        jacT<N> P; P.X = x; P.Y = y; fe_set_one(P.Z);
        jac_dbl<N, VtOps>(P);
        feT<N> zi = fe_inv(P.Z);
        fe_sqr(zi, zi);
        fe_mul(r, P.X, zi);
    } break;
    case 9: case 10: {                    // fe_reduce on a raw 2N-limb value: a = low half, b = high half
        uint32_t w[2 * N];
        for (int i = 0; i < N; ++i) { w[i] = x.v[i]; w[N + i] = y.v[i]; }
        if (op == 9) fe_reduce<1>(r, w); else fe_reduce<3>(r, w);
    } break;
    case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27: case 28: {
        // the lazy-limb forms of bign_fe29.hpp (29 / 28 / 27-bit limbs on the three curves)
        lzT<N> a29, b29, r29, t29;
        f29_from_words(a29, x);
        f29_from_words(b29, y);
        switch (op) {
        case 20: f29_mul(r29, a29, b29); break;
        case 21: f29_sqr(r29, a29); break;
        case 22: f29_mul<3>(r29, a29, b29); break;
        case 23: f29_sqr<8>(r29, a29); break;
        case 24: f29_sub(r29, a29, b29); break;                                          // lazy, straight out
        case 25: f29_sub(t29, a29, b29); f29_add(r29, a29, b29); f29_mul<3>(r29, t29, r29); break;   // 3 (a-b)(a+b)
        case 26: f29_sub(t29, a29, b29); f29_sub(t29, t29, b29); f29_sub(t29, t29, b29); f29_carry(t29);
                 f29_neg(r29, b29); f29_mul<4>(r29, t29, r29); break;                    // 4 (a - 3b)(-b)
        case 27: f29_sub(t29, a29, b29); f29_sqr<8>(r29, t29); break;                     // 8 (a-b)^2
        default: f29_add(t29, a29, b29); f29_sub(r29, a29, b29); f29_mul<2>(r29, r29, t29);
                 f29_sub(r29, r29, a29); f29_sub(r29, r29, a29); f29_sub(r29, r29, a29); break;  // 2(a-b)(a+b) - 3a
        }
        f29_to_words(r, r29);
    } break;
    case 29: case 30: case 31: case 32: {
        // point doubling / addition in the lazy-limb forms: affine x of 2P (29: one lane, 31: quad) / of 2P + P (30, 32)
        feT<N> X, Z;
        if (op <= 30) {
            ljacT<N> T;
            f29_from_words(T.X, x);
            f29_from_words(T.Y, y);
            f29_set_one(T.Z);
            laffT<N> E;
            E.x = T.X; E.y = T.Y;
            jac29_dbl(T);
            if (op == 30) jac29_madd(T, E);
            f29_to_words(X, T.X);
            f29_to_words(Z, T.Z);
        } else {
            // every quad of the launch works on the point of its FIRST lane (the test repeats each point four times)
            lqjacT<N> T;
            f29_from_words(T.X, x);
            f29_from_words(T.Y, y);
            f29_set_one(T.Z);
            f29_set_one(T.D);
            lqentT<N> E;
            E.X = T.X; E.Y = T.Y; E.Z = T.Z; E.ZZ = T.Z;
            quad29_dbl(T, (uint32_t)threadIdx.x & 3u);
            if (op == 32) quad29_add(T, E, (uint32_t)threadIdx.x & 3u);
            f29_to_words(X, T.X);
            f29_to_words(Z, T.Z);
        }
        feT<N> zi = fe_inv(Z);
        fe_sqr(zi, zi);
        fe_mul(r, X, zi);
    } break;
    default: {
        jacT<N> P; P.X = x; P.Y = y; fe_set_one(P.Z);
        jac_dbl(P);
        feT<N> zi = fe_inv(P.Z);
        fe_sqr(zi, zi);
        fe_mul(r, P.X, zi);
    } break;
    }
    fe_canon(r, r);
    for (int i = 0; i < N; ++i) out[N * idx + i] = r.v[i];
}

#endif   // BEE2HIP_EXPERIMENTS

// ------------------------------------------------------------------ host side ---
struct BignDevice {
    uint4 *gtab8[3] = {nullptr, nullptr, nullptr};     // 8-bit seed table per curve (index N/4 - 2): 4N x 256 affine points
    uint4 *gtab[3] = {nullptr, nullptr, nullptr};      // 16-bit comb table per curve, built from the seed table
    uint4 *gtab6[3] = {nullptr, nullptr, nullptr};     // signed 6-bit windows (signing side, one lane per scalar)
    uint4 *gtab7[3] = {nullptr, nullptr, nullptr};     // signed 7-bit windows (signing side, LDS look-up kernel: 256-bit curve)
    uint4 *gtabw8[3] = {nullptr, nullptr, nullptr};    // signed 8-bit windows (the same kernel with 16 copies of the row and 16-octet reads)
};
static BignDevice g_bign[64];
static std::mutex g_bign_mu;          // table construction is per device, shared by threads

// the seed table alone (0.5 / 1.7 / 4 MiB): all the signing / key generation path needs (bign_sign_kernels.hip).
// Caller holds g_bign_mu.
template <int N>
static err_t bign_table8_locked(uint4 **out, hipStream_t st)
{
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return ERR_BAD_INPUT;
    uint4 *&slot = g_bign[dev].gtab8[N / 4 - 2];
    if (!slot) {
        const size_t pt = 8 * N;                                      // bytes per affine point
        uint4 *t8 = nullptr;
        if (hipMalloc((void **)&t8, (size_t)4 * N * GT8_ENTRIES * pt) != hipSuccess) { (void)hipGetLastError(); return ERR_OUTOFMEMORY; }
        hipLaunchKernelGGL(bign_gtable_kernel<N>, dim3(4 * N * GT8_ENTRIES / 64), dim3(64), 0, st, t8);
        B2H_TRY(hipGetLastError());
        B2H_TRY(hipStreamSynchronize(st));
        slot = t8;
    }
    *out = slot;
    return ERR_OK;
}
template <int N>
static err_t bign_table8(const uint32_t **out, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_bign_mu);
    uint4 *t = nullptr;
    const err_t code = bign_table8_locked<N>(&t, st);
    *out = reinterpret_cast<const uint32_t *>(t);
    return code;
}

// seed table and the signed 6-bit table made from it (88 / 200 / 352 KiB)
template <int N>
static err_t bign_table6(const uint32_t **out8, const uint32_t **out6, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_bign_mu);
    uint4 *t8 = nullptr;
    err_t code = bign_table8_locked<N>(&t8, st);
    if (code != ERR_OK) return code;
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    uint4 *&slot = g_bign[dev].gtab6[N / 4 - 2];
    if (!slot) {
        uint4 *t6 = nullptr;
        const size_t entries = (size_t)Win6<N>::W * 32;
        if (hipMalloc((void **)&t6, entries * 8 * N) != hipSuccess) { (void)hipGetLastError(); return ERR_OUTOFMEMORY; }
        hipLaunchKernelGGL(bign_gtable6_kernel<N>, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, st, (const uint4 *)t8, t6);
        B2H_TRY(hipGetLastError());
        B2H_TRY(hipStreamSynchronize(st));
        slot = t6;
    }
    *out8 = reinterpret_cast<const uint32_t *>(t8);
    *out6 = reinterpret_cast<const uint32_t *>(slot);
    return ERR_OK;
}

// the signed 7- / 8-bit window tables (148 / 264 KiB on the 256-bit curve), made from the seed table like the 6-bit one
template <int N, int WB>
static err_t bign_tablew(const uint32_t **outw, hipStream_t st)
{
    static_assert(WB == 7 || WB == 8, "window tables of the LDS look-up kernel");
    std::lock_guard<std::mutex> lk(g_bign_mu);
    uint4 *t8 = nullptr;
    err_t code = bign_table8_locked<N>(&t8, st);
    if (code != ERR_OK) return code;
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    uint4 *&slot = (WB == 7 ? g_bign[dev].gtab7 : g_bign[dev].gtabw8)[N / 4 - 2];
    if (!slot) {
        uint4 *tw = nullptr;
        const size_t entries = (size_t)WinW<N, WB>::W * WinW<N, WB>::ENT;
        if (hipMalloc((void **)&tw, entries * 8 * N) != hipSuccess) { (void)hipGetLastError(); return ERR_OUTOFMEMORY; }
        hipLaunchKernelGGL((bign_gtable6_kernel<N, WB>), dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, st, (const uint4 *)t8, tw);
        B2H_TRY(hipGetLastError());
        B2H_TRY(hipStreamSynchronize(st));
        slot = tw;
    }
    *outw = reinterpret_cast<const uint32_t *>(slot);
    return ERR_OK;
}

template <int N>
static err_t bign_table(uint4 **out, hipStream_t st)
{
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return ERR_BAD_INPUT;
    uint4 *&slot = g_bign[dev].gtab[N / 4 - 2];
    if (!slot) {
        const size_t pt = 8 * N;
        uint4 *t8 = nullptr;
        const err_t code = bign_table8_locked<N>(&t8, st);
        if (code != ERR_OK) return code;
        uint4 *t16 = nullptr;
        if (hipMalloc((void **)&t16, (size_t)2 * N * 65536 * pt) != hipSuccess) { (void)hipGetLastError(); return ERR_OUTOFMEMORY; }
        hipLaunchKernelGGL(bign_gtable16_kernel<N>, dim3(2 * N * 65536 / 256), dim3(256), 0, st, (const uint4 *)t8, t16);
        B2H_TRY(hipGetLastError());
        B2H_TRY(hipStreamSynchronize(st));
        slot = t16;
    }
    *out = slot;
    return ERR_OK;
}

template <int N>
static err_t bign_scratch(hipStream_t st, size_t n, VerifyScratch &S)
{
    const size_t n_pad = (n + 63) & ~(size_t)63;
    const size_t words = n_pad * (1 + N + (N / 2 + 1) + 8 * 2 * N + 14 * N + N);
    void *base = nullptr;
    err_t code = scratch_for_stream(st, N / 4 - 2 + 4, words * 4, &base);
    if (code != ERR_OK) return code;
    uint32_t *p = (uint32_t *)base;
    S.n_pad = n_pad;
    S.status = p; p += n_pad;
    S.u = p; p += (size_t)N * n_pad;
    S.w = p; p += (size_t)(N / 2 + 1) * n_pad;
    S.qtab = p; p += (size_t)8 * 2 * N * n_pad;
    S.qz = p; p += (size_t)14 * N * n_pad;
    S.rx = p;
    return ERR_OK;
}

static int g_inv_lanes_log2 = 0;                     // lanes of bign_inv_kernel (log2; 0 = by curve): A/B, tune 23
void set_inv_lanes(int v) { g_inv_lanes_log2 = v; }
static int g_verify_path = 0, g_verify_lanes = 0;
void set_verify_path(int v) { g_verify_path = v & 15; g_verify_lanes = v >> 4; }   // 0x43: quads at every size
static int g_verify_split = 0;                       // 0 = by size, 1 = never, 2 / 3 / 4 = that many parts (A/B)
void set_verify_split(int v) { g_verify_split = v; }

// EXPERIMENT (round 3, off by default): a big batch as PARTS on separate streams, each part its own prep -> main -> slow ->
// inv -> tail chain, part p + 1 starting when part p's prep is through, so that the phases that leave the SIMDs half empty
// -- prep (two wavefronts per SIMD around one serial inversion), inv (512 wavefronts on 1024 SIMDs), tail (LDS-bound) --
// run beside another part's main kernel instead of alone.  It does not pay (profiles/r03_verify_split.txt).  The parts share the scratch (disjoint index ranges of the same
// signature-fastest arrays) and the comb table; the caller's stream is forked and joined with events.
struct AuxStreams {
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    int dev = -1;
    err_t get()
    {
        int cur = 0;
        B2H_TRY(hipGetDevice(&cur));
        if (s[0] && cur == dev) return ERR_OK;
        for (hipStream_t &x : s) { if (x) (void)hipStreamDestroy(x); x = nullptr; }
        for (hipStream_t &x : s) B2H_TRY(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        dev = cur;
        return ERR_OK;
    }
};
// multiply-adds in pairs in the verification MAIN kernel: -1 = by curve and batch size (launch_bign_verify_t), 0 never, else always
// (A/B: bee2hip_internal_tune 19)
static int g_verify_pairs = -1;
#ifdef BEE2HIP_EXPERIMENTS
void set_verify_pairs(int v) { g_verify_pairs = v; }
#endif
static thread_local AuxStreams t_aux;

template <int N>
static err_t launch_bign_verify_t(const uint8_t *oid_der, size_t oid_len, const void *d_hashes_all,
                                  const void *d_sigs_all, const void *d_pubkeys_all, size_t n, void *d_codes_all,
                                  hipStream_t st)
{
    uint4 *gtab = nullptr;
    err_t code;
    {
        std::lock_guard<std::mutex> lk(g_bign_mu);     // table construction happens once per device
        code = bign_table<N>(&gtab, st);
    }
    if (code != ERR_OK) return code;
    VerifyScratch S_all;
    code = bign_scratch<N>(st, n, S_all);
    if (code != ERR_OK) return code;
    OidArg oid;
    code = make_oid_arg(oid, oid_der, oid_len, st);
    if (code != ERR_OK) return code;
    const size_t n_total = n;
    constexpr size_t NO_ = 4 * N;
    // one chain over the signatures [off, off + cnt) on stream `st`; ev_prep (may be null) is recorded behind the prep kernel
    const auto run_range = [&, gtab, oid](size_t off, size_t cnt, hipStream_t st, hipEvent_t ev_prep) -> err_t {
    VerifyScratch S = S_all;
    S.status += off; S.u += off; S.w += off; S.qtab += off; S.qz += off; S.rx += off;
    const size_t n = cnt;
    const uint8_t *d_hashes = (const uint8_t *)d_hashes_all + NO_ * off;
    const uint8_t *d_sigs = (const uint8_t *)d_sigs_all + (NO_ + NO_ / 2) * off;
    const uint8_t *d_pubkeys = (const uint8_t *)d_pubkeys_all + 2 * NO_ * off;
    uint32_t *d_codes = (uint32_t *)d_codes_all + off;
    err_t code = ERR_OK;
    const unsigned g256 = (unsigned)((n + 255) / 256), g64 = (unsigned)((n + 63) / 64);
    const size_t sp = n_total >= ((size_t)1 << 18) ? 2 : 1;     // signatures per lane in prep (shared inversion)
    const size_t plan = (n + sp - 1) / sp;
    // Which kernels walk the scalar multiplication (g_verify_path: 0 by size, 1 always the 32-bit
    // kernels, 2 the 29-bit main kernel, 3 the quad / pair kernel, + 16 x lanes to force quads (0x43), pairs (0x23) or quads with a helper quad (0x83; 0x93 / 0xA3 with 64 / 256 lanes per block)
    // -- tests and A/B):
    //   <= 2^14 signatures: one signature per quad, 29-bit limbs (prep + main in one kernel, no table inversion)
    //   <= 2^15           : one signature per pair of lanes, same kernel (256-bit curve; the wider ones use quads up
    //                       to 2^14 and the 32-bit kernels above: 28- / 27-bit limbs, LZ<N>)
    //   <= 2^18           : one lane per signature, 29-bit limbs, the multiplication one asm block (bign_fe29_asm.inc; round 6: it
    //                       beats the 32-bit kernels at four wavefronts per SIMD too: 1 662 against 1 768 us at 2^18)
    //   above             : one lane per signature, 32-bit limbs (the throughput form)
    int path = 1;
    if constexpr (N == 8) path = g_verify_path ? g_verify_path : n <= ((size_t)1 << 15) ? 3 : n <= ((size_t)1 << 18) ? 2 : 1;   // (round 6: the 29-bit kernel's multiplication is one asm block now -- it beats the 32-bit kernels up to 2^18: profiles/r06_f29_asm_ab.txt)
    // wider curves, quads only: up to 2^14 signatures they leave one wavefront per SIMD; up to 2^15 two, which costs twice
    // the time (2.4 / 4.7 ms) and still beats the one-lane kernels' latency floor (3.4 / 6.7 ms, profiles/r03_verify_wide.txt)
    else path = g_verify_path == 1 || g_verify_path == 3 ? g_verify_path : n <= ((size_t)1 << 15) ? 3 : 1;
    if (path == 3) {
        const auto launch = [&](auto kern, unsigned wg, unsigned lanes) -> err_t {
            const unsigned ns = wg / lanes;
            const size_t lds = (size_t)8 * 4 * LZ<N>::L * ns * 4;
            if (lds > 48 * 1024) B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), lds));   // once per (device, kernel)
            hipLaunchKernelGGL(kern, dim3((unsigned)((n + ns - 1) / ns)), dim3(wg), lds, st, (const uint8_t *)d_hashes,
                               (const uint8_t *)d_sigs, (const uint8_t *)d_pubkeys, n, S, (const uint4 *)gtab);
            return ERR_OK;
        };
        int lanes = 4;
        if constexpr (N == 8) lanes = g_verify_lanes == 2 ? 2 : g_verify_lanes ? 4 : n <= ((size_t)1 << 14) ? 4 : 2;
        if constexpr (N == 8) {
            if (lanes == 2) code = launch(bign_quad29_kernel<8, 256, 2>, 256, 2);
        }
        if (lanes != 2) {
            // up to 2^13 signatures a helper quad per signature takes the comb of u off the main quad's chain (x1.05, x1.10
            // at 2^13).  Four-wave blocks, because wavefronts of a block share a CU's instruction fetches and this
            // kernel is long; around 2^12 one-wave blocks spread over all CUs win (tools/ab/verify_helper_ab.py).
            const bool helper = g_verify_lanes >= 8 || (g_verify_lanes == 0 && n <= ((size_t)1 << 13));
            const bool one_wave = g_verify_lanes == 9 || (g_verify_lanes != 10 && n > ((size_t)3 << 10) && n <= ((size_t)1 << 12));
            if (helper && one_wave) code = launch(bign_quad29_kernel<N, 64, 8>, 64, 8);
            else if (helper) code = launch(bign_quad29_kernel<N, 256, 8>, 256, 8);
#ifdef BEE2HIP_EXPERIMENTS      // reachable only with the helper quad switched off (tune 2): A/B record
            else if (n <= ((size_t)1 << 13)) code = launch(bign_quad29_kernel<N, 64, 4>, 64, 4);
#endif
            else code = launch(bign_quad29_kernel<N, 256, 4>, 256, 4);
        }
        if (code != ERR_OK) return code;
    }
    if (path != 3) {
        // The MAIN kernel takes its multiply-adds in pairs (VtOpsP, bign_dev.hpp mac2) where that pays -- measured, tools/ab/verify_pairs_ab.py,
        // profiles/r04_mad_pairs_vt.txt: the 384- / 512-bit curves up to 2^16 signatures (ONE wavefront per SIMD: 2.85 -> 2.14 ms and
        // 6.26 -> 4.82 ms at 2^16), the 256-bit curve from 2^18 on (four wavefronts: +0.8 % at 2^18, +1.7 % at 2^19); at two wavefronts
        // per SIMD the paired form loses 2-5 % on every curve, and points / prep do not care.  g_verify_pairs: -1 by size, 0 never, else always (A/B).
        const bool pair_main = g_verify_pairs >= 0 ? g_verify_pairs != 0
                             : N == 8 ? n >= ((size_t)1 << 18) : n <= ((size_t)1 << 16);
        if constexpr (N != 8)
            hipLaunchKernelGGL((bign_points_kernel<N>), dim3(g256), dim3(256), 0, st, (const uint8_t *)d_hashes,
                               (const uint8_t *)d_sigs, (const uint8_t *)d_pubkeys, n, S);
        hipLaunchKernelGGL((bign_prep_kernel<N>), dim3((unsigned)((plan + 255) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_hashes, (const uint8_t *)d_sigs, (const uint8_t *)d_pubkeys, n, plan,
                           (int)sp, S);
        if (ev_prep) B2H_TRY(hipEventRecord(ev_prep, st));
        if constexpr (N == 8) {
            if (path == 2) {
                const unsigned wg = n <= ((size_t)1 << 14) ? 64u : 256u;
                hipLaunchKernelGGL(bign_main29_kernel, dim3((unsigned)((n + wg - 1) / wg)), dim3(wg), 0, st, n, S,
                                   (const uint4 *)gtab);
            }
        }
        if (path != 2) {
            bool main_done = false;
            if (pair_main) {
                hipLaunchKernelGGL((bign_main_kernel<N, VtOpsP>), dim3(g256), dim3(256), 0, st, n, S, (const uint4 *)gtab);
                main_done = true;
            }
            if (!main_done) hipLaunchKernelGGL((bign_main_kernel<N, VtOps>), dim3(g256), dim3(256), 0, st, n, S, (const uint4 *)gtab);
        }
    }
    hipLaunchKernelGGL(bign_slow_kernel<N>, dim3(g64), dim3(64), 0, st, (const uint8_t *)d_sigs,
                       (const uint8_t *)d_pubkeys, n, S, (size_t)(2 * NO_));
    // signatures per inversion.  Each lane runs one chain (division steps, then 5 multiplications per
    // signature) and a lone wavefront issues at about a third of a SIMD's rate, so fewer, longer lanes cost
    // little until the lanes no longer cover the SIMDs: measured best at 2^18 signatures K = 8 on the 256-bit
    // curve (72 us; K = 2: 105 us) and K = 4 on the wider ones (profiles/r01_bign_ab_inv.txt)
    // (round 4, tools/ab/inv_lanes_ab.py: with the division-step inversion the optimum is flat; up to 2^17 signatures on the 256-bit curve 2^16
    //  lanes -- one wavefront per SIMD, two signatures each -- are 4-7 % ahead of 2^15; profiles/r04_inv_lanes_ab.txt)
    const size_t inv_lanes = g_inv_lanes_log2 > 0 ? (size_t)1 << g_inv_lanes_log2 : N == 8 && n > ((size_t)1 << 17) ? 32768 : 65536;
    const size_t k_inv = std::min<size_t>(16, std::max<size_t>(1, n / inv_lanes));
    const size_t lanes = (n + k_inv - 1) / k_inv;
    hipLaunchKernelGGL(bign_inv_kernel<N>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, n, lanes,
                       (int)k_inv, S);
    constexpr size_t row_bytes = (2 * N + 1) * 4;
    if (N == 8 && n >= 65536) {
        auto kern = bign_tail_kernel<N, BeltTabTwoP, 1024>;
        const size_t lds = BeltTabTwo::kBytes + 1024 * row_bytes;
        B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), lds, st, (const uint8_t *)d_hashes,
                           (const uint8_t *)d_sigs, n, S, oid, (uint32_t *)d_codes);
    } else {
        hipLaunchKernelGGL((bign_tail_kernel<N, BeltTabSmall, 64>), dim3(g64), dim3(64),
                           BeltTabSmall::kBytes + 64 * row_bytes, st, (const uint8_t *)d_hashes,
                           (const uint8_t *)d_sigs, n, S, oid, (uint32_t *)d_codes);
    }
    B2H_TRY(hipGetLastError());
    return ERR_OK;
    };   // run_range

    // parts: only the one-lane throughput kernels of a big batch
    size_t parts = 1;
    {
        const bool one_lane = N == 8 ? (g_verify_path == 1 || (g_verify_path == 0 && n_total > ((size_t)1 << 16)))
                                     : (g_verify_path == 1 || (g_verify_path == 0 && n_total > ((size_t)1 << 15)));
        // measured (profiles/r03_verify_split.txt): no gain -- 2 parts -1.5 % at 2^18, +-0 at 2^19 / 2^20, 3-4 parts -5..-30 %; the
        // default is therefore ONE part, and the parts stay reachable through bee2hip_internal_tune(8, parts) for the record
        if (one_lane) parts = g_verify_split >= 2 ? (size_t)g_verify_split : 1;
        if (parts > 4) parts = 4;
        if (n_total < parts * 4096) parts = 1;
    }
    if (parts == 1) return run_range(0, n_total, st, nullptr);
    code = t_aux.get();
    if (code != ERR_OK) return code;
    hipEvent_t ev[9];                               // fork, prep done x 4, join x 4
    for (hipEvent_t &e : ev) e = nullptr;
    const auto cleanup = [&]() { for (hipEvent_t &e : ev) if (e) (void)hipEventDestroy(e); };
    for (size_t i = 0; i < 1 + 2 * parts; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { cleanup(); return hip_fail(hipGetLastError(), "hipEventCreate"); }
    const size_t per = ((n_total / parts + 255) / 256) * 256;
    hipError_t he = hipEventRecord(ev[0], st);       // everything queued so far (inputs, the OID prefix) is before the fork
    for (size_t p = 0; p < parts && he == hipSuccess && code == ERR_OK; ++p) {
        const size_t off = p * per, cnt = p + 1 == parts ? n_total - off : per;
        const hipStream_t sp_ = p == 0 ? st : t_aux.s[p - 1];
        if (p) {
            he = hipStreamWaitEvent(sp_, ev[0], 0);
            if (he == hipSuccess) he = hipStreamWaitEvent(sp_, ev[p], 0);          // the previous part's prep is through
            if (he != hipSuccess) break;
        }
        code = run_range(off, cnt, sp_, ev[1 + p]);
        if (code == ERR_OK && p) he = hipEventRecord(ev[1 + parts + p], sp_);
    }
    for (size_t p = 1; p < parts && he == hipSuccess && code == ERR_OK; ++p) he = hipStreamWaitEvent(st, ev[1 + parts + p], 0);
    cleanup();                                      // destroying a pending event is allowed: it is released when it completes
    if (code != ERR_OK) return code;
    if (he != hipSuccess) return hip_fail(he, "verify parts");
    return ERR_OK;
}

// l = security level (128, 192, 256): hashes n*(l/4), sigs n*(3l/8), pubkeys n*(l/2) octets
err_t launch_bign_verify(size_t l, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                         const void *d_sigs, const void *d_pubkeys, size_t n, void *d_codes,
                         hipStream_t st)
{
    if (n == 0) return ERR_OK;
    if (l == 128) return launch_bign_verify_t<8>(oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, st);
    if (l == 192) return launch_bign_verify_t<12>(oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, st);
    if (l == 256) return launch_bign_verify_t<16>(oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, st);
    return ERR_BAD_PARAMS;
}

// ---- one signer: the host's share ----------------------------------------------------------------------------------
// Is Q a point of the curve, and the 2N + 1 points 2^(8 w) Q the table kernel starts from: a chain of l doublings, ~25 us on a
// host core (0.5 ms for a lone wavefront), made affine with one inversion.  NL = 64-bit limbs (N = 2 NL).
template <int NL>
static bool onekey_base_t(uint64_t crandall_c, const uint8_t *b_le, const uint8_t *pubkey, std::vector<uint8_t> &base)
{
    hostb::Curve<NL> E;                                 // (group law and field only: its tables of G are not needed)
    E.F.c = crandall_c;
    if (hostb::pubkey_val<NL>(E, b_le, pubkey) != ERR_OK) return false;
    constexpr int W = 4 * NL + 1;                       // windows of 8 bits over l + 1 bits
    const hostb::Field<NL> &F = E.F;
    hostb::Aff<NL> Q;
    for (int i = 0; i < NL; ++i) { Q.x.v[i] = hostp::ld64le(pubkey + 8 * i); Q.y.v[i] = hostp::ld64le(pubkey + 8 * NL + 8 * i); }
    std::vector<hostb::Jac<NL>> P(W);
    E.from_aff(P[0], Q, false);
    for (int w = 1; w < W; ++w) {
        P[w] = P[w - 1];
        for (int k = 0; k < 8; ++k) { hostb::Jac<NL> t; E.dbl(t, P[w]); P[w] = t; }
    }
    // to affine with ONE inversion (Z != 0: a point of prime order q > 2^(l + 8) is doubled)
    std::vector<hostb::Fe<NL>> pre(W);
    hostb::Fe<NL> acc = P[0].Z;
    pre[0] = acc;
    for (int w = 1; w < W; ++w) { F.mul(acc, acc, P[w].Z); pre[w] = acc; }
    if (hostb::Field<NL>::is_zero(acc)) return false;
    hostb::Fe<NL> inv;
    F.inv(inv, acc);
    base.resize((size_t)W * 16 * NL);
    for (int w = W - 1; w >= 0; --w) {
        hostb::Fe<NL> zi, zi2, x, y;
        if (w) { F.mul(zi, inv, pre[w - 1]); F.mul(inv, inv, P[w].Z); } else zi = inv;
        F.sqr(zi2, zi);
        F.mul(x, P[w].X, zi2);
        F.mul(zi2, zi2, zi);
        F.mul(y, P[w].Y, zi2);
        uint8_t *o = base.data() + (size_t)w * 16 * NL;
        for (int i = 0; i < NL; ++i) { hostp::st64le(o + 8 * i, x.v[i]); hostp::st64le(o + 8 * NL + 8 * i, y.v[i]); }
    }
    return true;
}
template <int N>
static bool onekey_base(const uint8_t *pubkey, std::vector<uint8_t> &base)
{
    if (N == 8) return onekey_base_t<4>(BIGN128_CRANDALL_C, k_bign128_b, pubkey, base);
    if (N == 12) return onekey_base_t<6>(BIGN192_CRANDALL_C, k_bign192_b, pubkey, base);
    return onekey_base_t<8>(BIGN256_CRANDALL_C, k_bign256_b, pubkey, base);
}

// ---- one signer: table cache and launcher --------------------------------------------------------------------------
// The comb table of a public key lives in device memory, keyed by (device, curve, key octets); the last KEYTAB_SLOTS keys per
// process are kept (a table is 278 KiB .. 1 MiB: the card holds as many as anybody wants, the limit only bounds the scan).
// A launcher holds a reference while it queues its kernels; the table is freed when the last reference goes, and hipFree
// waits for the device, so a kernel already queued never loses its table.
// (round 5) A slab holds at most KEYSLAB_KEYS tables, so ONE busy key pins at most that many (4.4 - 16 MiB), and the
// cache is bounded in BYTES as well (KEYTAB_BYTES_MAX of live slabs): before a new slab is allocated the least recently used
// entries go until there is room, and when hipMalloc still refuses, every entry nobody holds goes and the allocation is tried
// once more.  Keys that end up without a table are verified by the paths that need none (one signer: the general pipeline;
// several: the complete-formula kernel) -- slower, same verdicts, no error.
constexpr size_t KEYSLAB_KEYS = 16;
constexpr size_t KEYTAB_BYTES_MAX = (size_t)8 << 30;
static std::atomic<size_t> g_keyslab_bytes{0};
struct KeySlab {                       // tables of keys one call met for the first time, in one device allocation
    void *p = nullptr;
    size_t bytes = 0;
    ~KeySlab() { if (p) { (void)hipFree(p); g_keyslab_bytes.fetch_sub(bytes); } }
};
struct KeyTab {
    std::shared_ptr<KeySlab> slab;     // (freed with the last of its tables)
    const uint4 *tab = nullptr;        // (2N + 1) x 256 affine points
    const uint8_t *d_key = nullptr;    // the key itself in device memory (2 NO octets)
    uint4 *tab16 = nullptr;            // N x 65536 affine points (32 / 72 / 128 MiB): once `used` says the key is a busy one
    uint64_t stamp = 0, used = 0;      // signatures verified under the key so far
    ~KeyTab();
};
// at most KEYTAB16_MAX keys hold a 16-bit table at a time (<= 3 / 7 / 12 GiB per process): a key that becomes busy while they are
// all taken stays on its 8-bit table
constexpr int KEYTAB16_MAX = 96;
static std::atomic<int> g_keytab16_live{0};
KeyTab::~KeyTab() { if (tab16) { (void)hipFree(tab16); g_keytab16_live.fetch_sub(1); } }
static size_t KEYTAB_SLOTS = 1024;                   // (0.3 - 1 GiB of 8-bit tables when full; tests: tune 21)
void set_onekey_slots(int v) { KEYTAB_SLOTS = v < 1 ? 1 : (size_t)v; }
static std::unordered_map<std::string, std::shared_ptr<KeyTab>> &g_keytabs = *new std::unordered_map<std::string, std::shared_ptr<KeyTab>>;   // (never destroyed: no hipFree behind the runtime's back at exit)
// (round 5) the key-table cache has a lock of its own: building a key's tables (host arithmetic, a kernel, a synchronise:
// 0.2-2 ms) holds it, and general verification -- which takes g_bign_mu to fetch G's table -- no longer waits behind that.  Never nested
// with g_bign_mu.
static std::mutex g_keytab_mu;
static uint64_t g_keytab_clock = 0;
static std::atomic<unsigned long long> g_keytab_builds{0};
unsigned long long bign_onekey_table_builds() { return g_keytab_builds.load(); }

// A key under which KEYTAB16_AFTER signatures have been verified gets the 16-bit table as well (0.3 / 1 / 2 ms to build -- the work
// of ~2^17 signatures -- against a quarter of the additions saved from then on): made from the 8-bit one like G's (bign_gtable16_kernel).
static int g_onekey_quads = -1;                      // four lanes per signature: -1 by batch size (<= 2^16), 0 never, 1 always (tests / A/B: tune 22)
void set_onekey_quads(int v) { g_onekey_quads = v; }
static int g_keytab16_log2 = -1;                     // -1: by curve (2^19 on the 256-bit curve, 2^20 on the wider ones); tests / A/B: tune 20
void set_onekey_tab16(int v) { g_keytab16_log2 = v; }
template <int N>
static err_t bign_key_table16_locked(KeyTab &k, hipStream_t st)
{
    const size_t pt = 8 * N;
    uint4 *t16 = nullptr;
    if (g_keytab16_live.load() >= KEYTAB16_MAX) return ERR_OK;                                                        // enough busy keys already
    if (hipMalloc((void **)&t16, (size_t)N * 65536 * pt) != hipSuccess) { (void)hipGetLastError(); return ERR_OK; }   // no room: the 8-bit table serves
    hipLaunchKernelGGL(bign_gtable16_kernel<N>, dim3(N * 65536 / 256), dim3(256), 0, st, k.tab, t16);
    hipError_t e16 = hipGetLastError();
    if (e16 == hipSuccess) e16 = hipStreamSynchronize(st);
    if (e16 != hipSuccess) { (void)hipFree(t16); return hip_fail(e16, "bign_gtable16_kernel (key)"); }
    k.tab16 = t16;
    g_keytab16_live.fetch_add(1);
    return ERR_OK;
}

// The tables of nkeys keys (host memory, 2 NO octets each): out[k] = the cached entry, or null for a key that is not a point of
// the curve.  The keys this process has not met (or has dropped) are built TOGETHER: their starting points on host threads (~35 us
// a key), one allocation [tables | key, starting points per key], one upload, one launch of bign_ktable_kernel (0.17 ms for one key,
// ~0.3 ms for 64: lone wavefronts fill the chip), one synchronisation.  n = signatures of this call (counted towards the 16-bit
// table of a key when want16).
template <int N>
static err_t bign_key_tables(std::vector<std::shared_ptr<KeyTab>> &out, const uint8_t *keys, size_t nkeys, size_t n, bool want16,
                             hipStream_t st)
{
    constexpr size_t NO = 4 * N, pt = 8 * N;
    constexpr size_t TAB = (size_t)(2 * N + 1) * GT8_ENTRIES * pt, AUX = 2 * NO + (size_t)(2 * N + 1) * pt;     // table; key + starting points
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    const int lg = g_keytab16_log2 >= 0 ? g_keytab16_log2 : N == 8 ? 19 : 20;
    const uint64_t after = lg >= 63 ? ~(uint64_t)0 : (uint64_t)1 << lg;
    out.assign(nkeys, nullptr);
    const auto id_of = [&](const uint8_t *key) {
        std::string id(2 + 2 * NO, '\0');
        id[0] = (char)dev; id[1] = (char)N;
        memcpy(&id[2], key, 2 * NO);
        return id;
    };
    //  entries that leave the cache are parked here and released AFTER the lock is dropped (declared before the lock:
    // destroyed after it): their destructors call hipFree, a device-wide wait, and every other key-table user would stall behind it
    std::vector<std::shared_ptr<KeyTab>> dropped;
    std::unique_lock<std::mutex> lk(g_keytab_mu);
    std::vector<size_t> miss;                          // first occurrence of every key the cache does not hold
    std::unordered_map<std::string, size_t> miss_at;   // id -> position in miss
    std::vector<size_t> dup_of(nkeys, (size_t)-1);     // later occurrences of a missing key
    for (size_t k = 0; k < nkeys; ++k) {
        std::string id = id_of(keys + 2 * NO * k);
        const auto it = g_keytabs.find(id);
        if (it != g_keytabs.end()) {
            KeyTab *t = it->second.get();
            t->stamp = ++g_keytab_clock;
            t->used += n;
            if (want16 && !t->tab16 && t->used >= after) { const err_t c = bign_key_table16_locked<N>(*t, st); if (c != ERR_OK) return c; }
            out[k] = it->second;
            continue;
        }
        const auto m = miss_at.find(id);
        if (m != miss_at.end()) { dup_of[k] = miss[m->second]; continue; }
        miss_at.emplace(std::move(id), miss.size());
        miss.push_back(k);
    }
    if (!miss.empty()) {
        {
            // a key this process has not met needs an allocation, an upload and a synchronisation: none of that is legal on a stream
            // that is being captured (and would kill the capture).  Say so instead: verify once under the key outside the capture.
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
                return hip_fail(hipErrorStreamCaptureUnsupported, "key-table miss under stream capture: verify once under each key before capturing");
            }
            (void)hipGetLastError();
        }
        // on the curve?  then the 2N + 1 starting points -- host threads when there are many keys
        std::vector<std::vector<uint8_t>> base(miss.size());
        std::vector<char> on_curve(miss.size(), 0);
        std::atomic<int> threw{0};                     // (an exception must not leave a worker thread: bad_alloc in a resize)
        const auto work = [&](size_t from, size_t to) {
            try {
                for (size_t i = from; i < to; ++i) on_curve[i] = onekey_base<N>(keys + 2 * NO * miss[i], base[i]) ? 1 : 0;
            } catch (...) { threw.store(1); }
        };
        const size_t nthr = std::min<size_t>(16, miss.size() / 4);
        if (nthr >= 2) {
            std::vector<std::thread> th;
            const size_t per = (miss.size() + nthr - 1) / nthr;
            size_t started = 0;
            try {
                for (; started < nthr; ++started)
                    th.emplace_back(work, std::min(miss.size(), started * per), std::min(miss.size(), (started + 1) * per));
            } catch (...) {                            // no more threads to be had: the rest on this one
                work(std::min(miss.size(), started * per), miss.size());
            }
            for (auto &t : th) t.join();
        } else work(0, miss.size());
        if (threw.load()) return ERR_OUTOFMEMORY;
        std::vector<size_t> good;
        for (size_t i = 0; i < miss.size(); ++i) if (on_curve[i]) good.push_back(i);
        const auto evict_lru = [&]() -> bool {                  // the least recently used entry leaves the cache (freed when its last user lets go)
            if (g_keytabs.empty()) return false;
            auto old = g_keytabs.begin();
            for (auto it = g_keytabs.begin(); it != g_keytabs.end(); ++it) if (it->second->stamp < old->second->stamp) old = it;
            dropped.push_back(std::move(old->second));
            g_keytabs.erase(old);
            return true;
        };
        // The BYTE bound evicts by slab: a slab's memory returns only when ALL of its (<= 16) tables are gone and nobody
        // holds one, so the unit that leaves is the idle slab whose most recent use is the oldest; when no idle slab is left the loop
        // stops -- it never empties the cache without freeing a byte.  Returns the bytes that will be freed once `dropped` is released.
        size_t pending_free = 0;
        const auto evict_idle_slab = [&]() -> size_t {
            struct Seen { uint64_t newest = 0; bool busy = false; };
            std::unordered_map<const KeySlab *, Seen> slabs;
            for (const auto &kv : g_keytabs) {
                Seen &e = slabs[kv.second->slab.get()];
                e.newest = std::max(e.newest, kv.second->stamp);
                e.busy |= kv.second.use_count() != 1;       // a launcher (or a captured graph) holds this table
            }
            const KeySlab *victim = nullptr;
            uint64_t oldest = ~(uint64_t)0;
            for (const auto &kv : slabs) if (kv.first && !kv.second.busy && kv.second.newest < oldest) { oldest = kv.second.newest; victim = kv.first; }
            if (!victim) return 0;
            const size_t bytes = victim->bytes;
            for (auto it = g_keytabs.begin(); it != g_keytabs.end();)
                if (it->second->slab.get() == victim) { dropped.push_back(std::move(it->second)); it = g_keytabs.erase(it); } else ++it;
            return bytes;
        };
        for (size_t g0 = 0; g0 < good.size(); g0 += KEYSLAB_KEYS) {
            const size_t M = std::min(KEYSLAB_KEYS, good.size() - g0);
            const size_t need = M * (TAB + AUX);
            while (g_keyslab_bytes.load() - std::min(pending_free, g_keyslab_bytes.load()) + need > KEYTAB_BYTES_MAX) {
                const size_t freed = evict_idle_slab();
                if (!freed) break;
                pending_free += freed;
            }
            auto slab = std::make_shared<KeySlab>();
            if (hipMalloc(&slab->p, need) != hipSuccess) {
                (void)hipGetLastError();
                slab->p = nullptr;
                // no room on the device: everything nobody is using right now goes -- released HERE, with the lock dropped, because the
                // retry needs the memory back now -- then one more try.  (Another thread may cache one of this call's keys meanwhile:
                // the assignment below then replaces its entry, which is harmless.)
                for (auto it = g_keytabs.begin(); it != g_keytabs.end();)
                    if (it->second.use_count() == 1) { dropped.push_back(std::move(it->second)); it = g_keytabs.erase(it); } else ++it;
                lk.unlock();
                dropped.clear();
                pending_free = 0;
                lk.lock();
                if (hipMalloc(&slab->p, need) != hipSuccess) {
                    (void)hipGetLastError();
                    slab->p = nullptr;
                    break;                                      // the remaining keys stay without a table: out[] = null, the callers' table-free paths
                }
            }
            slab->bytes = need;
            g_keyslab_bytes.fetch_add(need);
            uint8_t *d_tabs = reinterpret_cast<uint8_t *>(slab->p), *d_aux = d_tabs + M * TAB;
            std::vector<uint8_t> aux(M * AUX);
            for (size_t j = 0; j < M; ++j) {
                memcpy(aux.data() + j * AUX, keys + 2 * NO * miss[good[g0 + j]], 2 * NO);
                memcpy(aux.data() + j * AUX + 2 * NO, base[good[g0 + j]].data(), AUX - 2 * NO);
            }
            B2H_TRY(hipMemcpyAsync(d_aux, aux.data(), aux.size(), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(bign_ktable_kernel<N>, dim3((unsigned)(((size_t)(2 * N + 1) * GT8_ENTRIES + 63) / 64), (unsigned)M), dim3(64), 0, st,
                               reinterpret_cast<const uint4 *>(d_aux + 2 * NO), reinterpret_cast<uint4 *>(d_tabs), AUX / 16, TAB / 16);
            B2H_TRY(hipGetLastError());
            B2H_TRY(hipStreamSynchronize(st));          // (aux goes out of scope; other streams may use the tables next)
            for (size_t j = 0; j < M; ++j) {
                auto t = std::make_shared<KeyTab>();
                t->slab = slab;
                t->tab = reinterpret_cast<const uint4 *>(d_tabs + j * TAB);
                t->d_key = d_aux + j * AUX;
                t->stamp = ++g_keytab_clock;
                t->used = n;
                g_keytab_builds.fetch_add(1);
                if (want16 && t->used >= after) { const err_t c = bign_key_table16_locked<N>(*t, st); if (c != ERR_OK) return c; }
                if (g_keytabs.size() >= KEYTAB_SLOTS) evict_lru();
                g_keytabs[id_of(keys + 2 * NO * miss[good[g0 + j]])] = t;
                out[miss[good[g0 + j]]] = t;
            }
        }
        for (size_t k = 0; k < nkeys; ++k) if (dup_of[k] != (size_t)-1) out[k] = out[dup_of[k]];
    }
    return ERR_OK;
}

// keys: nkeys public keys on the host; d_key_index: null = ONE key (nkeys = 1), else n x uint32 on the device
template <int N>
static err_t launch_bign_verify_onekey_t(const uint8_t *oid_der, size_t oid_len, const void *d_hashes, const void *d_sigs,
                                         const uint8_t *keys, size_t nkeys, const void *d_key_index, size_t n, void *d_codes,
                                         hipStream_t st)
{
    constexpr size_t NO = 4 * N;
    const bool keyed = d_key_index != nullptr;
    uint4 *gtab = nullptr;
    err_t code;
    {
        std::lock_guard<std::mutex> lk(g_bign_mu);
        code = bign_table<N>(&gtab, st);
    }
    if (code != ERR_OK) return code;
    std::vector<std::shared_ptr<KeyTab>> kts;
    // (keyed: every key is credited with its share of the batch -- the signers of a batch are taken to be about equally busy)
    code = bign_key_tables<N>(kts, keys, nkeys, keyed ? (n + nkeys - 1) / nkeys : n, true, st);
    if (code != ERR_OK) return code;
    {
        // A stream CAPTURE bakes the tables' addresses into the graph, and nothing tells the library when that graph dies: the tables
        // a capture refers to are pinned for the life of the process (a reference that is never dropped), so neither the LRU nor a
        // later call's eviction can free memory a replay will read.  Bounded by the distinct keys ever captured.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
            std::lock_guard<std::mutex> lk(g_keytab_mu);
            static std::vector<std::shared_ptr<KeyTab>> &pinned = *new std::vector<std::shared_ptr<KeyTab>>;
            for (const auto &t : kts)
                if (t && std::find(pinned.begin(), pinned.end(), t) == pinned.end()) pinned.push_back(t);
        } else (void)hipGetLastError();
    }
    const uint4 *tab16 = nullptr;
    const uint8_t *d_keys = nullptr;
    const uint4 *const *d_tabs = nullptr;
    if (!keyed) {
        if (!kts[0]) return ERR_KEY_NOT_ON_CURVE;                  // (the caller's general path)
        {
            std::lock_guard<std::mutex> lk(g_keytab_mu);             // another thread may be giving the key its 16-bit table
            tab16 = kts[0]->tab16;
        }
        d_keys = kts[0]->d_key;
    } else {
        // the tables' addresses (a key off the curve has none: null) and the keys go up in one block
        const size_t koff = (16 * nkeys + 15) & ~(size_t)15;       // (the kernels read keys with 16-octet loads)
        std::vector<uint8_t> blk(koff + 2 * NO * nkeys);
        uint64_t *ptrs = reinterpret_cast<uint64_t *>(blk.data());
        {
            std::lock_guard<std::mutex> lk(g_keytab_mu);             // (another thread may be giving a key its 16-bit table)
            for (size_t k = 0; k < nkeys; ++k) {
                ptrs[k] = kts[k] ? (uint64_t)(uintptr_t)kts[k]->tab : 0;
                ptrs[nkeys + k] = kts[k] ? (uint64_t)(uintptr_t)kts[k]->tab16 : 0;
            }
        }
        memcpy(blk.data() + koff, keys, 2 * NO * nkeys);
        void *d_blk = nullptr;
        code = scratch_for_stream(st, 13, blk.size(), &d_blk);
        if (code != ERR_OK) return code;
        B2H_TRY(hipMemcpyAsync(d_blk, blk.data(), blk.size(), hipMemcpyHostToDevice, st));
        B2H_TRY(hipStreamSynchronize(st));                          // (blk is pageable and goes out of scope)
        d_tabs = reinterpret_cast<const uint4 *const *>(d_blk);
        d_keys = reinterpret_cast<const uint8_t *>(d_blk) + koff;
    }
    VerifyScratch S;
    code = bign_scratch<N>(st, n, S);
    if (code != ERR_OK) return code;
    OidArg oid;
    code = make_oid_arg(oid, oid_der, oid_len, st);
    if (code != ERR_OK) return code;
    const unsigned g256 = (unsigned)((n + 255) / 256), g64 = (unsigned)((n + 63) / 64);
    const uint8_t *dh = (const uint8_t *)d_hashes, *dsg = (const uint8_t *)d_sigs;
    // up to 2^16 signatures: four lanes per signature (a third of the latency; at 2^17 the one-lane kernels' two wavefronts per SIMD tie)
    const bool quad = g_onekey_quads > 0 || (g_onekey_quads < 0 && n <= ((size_t)1 << 16));
    const unsigned g4 = (unsigned)((4 * n + 255) / 256);
    if (quad && keyed)
        hipLaunchKernelGGL((bign_onekey4_kernel<N, VtOps, false, true>), dim3(g4), dim3(256), 0, st, dh, dsg, d_keys, n, S, (const uint4 *)gtab,
                           (const uint4 *)nullptr, (const uint4 *)nullptr, (const uint32_t *)d_key_index, d_tabs, (uint32_t)nkeys);
    else if (quad && tab16)
        hipLaunchKernelGGL((bign_onekey4_kernel<N, VtOps, true, false>), dim3(g4), dim3(256), 0, st, dh, dsg, d_keys, n, S, (const uint4 *)gtab,
                           kts[0]->tab, tab16, (const uint32_t *)nullptr, (const uint4 *const *)nullptr, 1u);
    else if (quad)
        hipLaunchKernelGGL((bign_onekey4_kernel<N, VtOps, false, false>), dim3(g4), dim3(256), 0, st, dh, dsg, d_keys, n, S, (const uint4 *)gtab,
                           kts[0]->tab, (const uint4 *)nullptr, (const uint32_t *)nullptr, (const uint4 *const *)nullptr, 1u);
    else if (keyed)
        hipLaunchKernelGGL((bign_onekey_kernel<N, VtOps, false, true>), dim3(g256), dim3(256), 0, st, dh, dsg, d_keys, n, S, (const uint4 *)gtab,
                           (const uint4 *)nullptr, (const uint4 *)nullptr, (const uint32_t *)d_key_index, d_tabs, (uint32_t)nkeys);
    else if (tab16)
        hipLaunchKernelGGL((bign_onekey_kernel<N, VtOps, true, false>), dim3(g256), dim3(256), 0, st, dh, dsg, d_keys, n, S, (const uint4 *)gtab,
                           kts[0]->tab, tab16, (const uint32_t *)nullptr, (const uint4 *const *)nullptr, 1u);
    else
        hipLaunchKernelGGL((bign_onekey_kernel<N, VtOps, false, false>), dim3(g256), dim3(256), 0, st, dh, dsg, d_keys, n, S, (const uint4 *)gtab,
                           kts[0]->tab, (const uint4 *)nullptr, (const uint32_t *)nullptr, (const uint4 *const *)nullptr, 1u);
    hipLaunchKernelGGL(bign_slow_kernel<N>, dim3(g64), dim3(64), 0, st, dsg, d_keys, n, S, keyed ? ~(size_t)0 : (size_t)0);
    // shared inversions and the hash tail: as launch_bign_verify_t
    // (round 4, tools/ab/inv_lanes_ab.py: with the division-step inversion the optimum is flat; up to 2^17 signatures on the 256-bit curve 2^16
    //  lanes -- one wavefront per SIMD, two signatures each -- are 4-7 % ahead of 2^15; profiles/r04_inv_lanes_ab.txt)
    const size_t inv_lanes = g_inv_lanes_log2 > 0 ? (size_t)1 << g_inv_lanes_log2 : N == 8 && n > ((size_t)1 << 17) ? 32768 : 65536;
    const size_t k_inv = std::min<size_t>(16, std::max<size_t>(1, n / inv_lanes));
    const size_t lanes = (n + k_inv - 1) / k_inv;
    hipLaunchKernelGGL(bign_inv_kernel<N>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, n, lanes, (int)k_inv, S);
    constexpr size_t row_bytes = (2 * N + 1) * 4;
    if (N == 8 && n >= 65536) {
        auto kern = bign_tail_kernel<N, BeltTabTwoP, 1024>;
        const size_t lds = BeltTabTwo::kBytes + 1024 * row_bytes;
        B2H_TRY(dyn_lds_once(reinterpret_cast<const void *>(kern), lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), lds, st, dh, dsg, n, S, oid, (uint32_t *)d_codes);
    } else {
        hipLaunchKernelGGL((bign_tail_kernel<N, BeltTabSmall, 64>), dim3(g64), dim3(64), BeltTabSmall::kBytes + 64 * row_bytes, st,
                           dh, dsg, n, S, oid, (uint32_t *)d_codes);
    }
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// n signatures under ONE public key of a standard curve; pubkey (2 NO octets) is HOST memory.  ERR_KEY_NOT_ON_CURVE: see common.hpp.
err_t launch_bign_verify_onekey(size_t l, const uint8_t *oid_der, size_t oid_len, const void *d_hashes, const void *d_sigs,
                                const uint8_t *pubkey, size_t n, void *d_codes, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    try {
        if (l == 128) return launch_bign_verify_onekey_t<8>(oid_der, oid_len, d_hashes, d_sigs, pubkey, 1, nullptr, n, d_codes, st);
        if (l == 192) return launch_bign_verify_onekey_t<12>(oid_der, oid_len, d_hashes, d_sigs, pubkey, 1, nullptr, n, d_codes, st);
        if (l == 256) return launch_bign_verify_onekey_t<16>(oid_der, oid_len, d_hashes, d_sigs, pubkey, 1, nullptr, n, d_codes, st);
    } catch (...) { return ERR_OUTOFMEMORY; }               // (bad_alloc, a thread that could not be started: nothing may leave the C ABI)
    return ERR_BAD_PARAMS;
}
// n signatures of nkeys signers (pubkeys: HOST, nkeys x 2 NO octets; d_key_index: n x uint32 on the device)
err_t launch_bign_verify_keyed(size_t l, const uint8_t *oid_der, size_t oid_len, const void *d_hashes, const void *d_sigs,
                               const uint8_t *pubkeys, size_t nkeys, const void *d_key_index, size_t n, void *d_codes, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    if (nkeys == 0 || !d_key_index) return ERR_BAD_INPUT;
    try {
        if (l == 128) return launch_bign_verify_onekey_t<8>(oid_der, oid_len, d_hashes, d_sigs, pubkeys, nkeys, d_key_index, n, d_codes, st);
        if (l == 192) return launch_bign_verify_onekey_t<12>(oid_der, oid_len, d_hashes, d_sigs, pubkeys, nkeys, d_key_index, n, d_codes, st);
        if (l == 256) return launch_bign_verify_onekey_t<16>(oid_der, oid_len, d_hashes, d_sigs, pubkeys, nkeys, d_key_index, n, d_codes, st);
    } catch (...) { return ERR_OUTOFMEMORY; }               // (bad_alloc, a thread that could not be started: nothing may leave the C ABI)
    return ERR_BAD_PARAMS;
}

// the fallback of the one-key entries (a key off the curve, a non-standard parameter set): the key n times, then the general path
__global__ __launch_bounds__(256)
void bign_replicate_key_kernel(const uint4 *__restrict__ key, uint4 *__restrict__ out, size_t quads_per_key, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = key[i % quads_per_key];
}
err_t launch_replicate_key(const void *d_key, size_t key_bytes, size_t n, void *d_out, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    const size_t q = key_bytes / 16, total = q * n;
    hipLaunchKernelGGL(bign_replicate_key_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint4 *)d_key, (uint4 *)d_out, q, total);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

// pubkeys n*(l/2) octets (16-byte aligned), codes n err_t
err_t launch_bign_pubkey_val(size_t l, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    const dim3 g((unsigned)((n + 255) / 256)), t(256);
    if (l == 128) hipLaunchKernelGGL(bign_pubkey_val_kernel<8>, g, t, 0, st, (const uint8_t *)d_pubkeys, n, (uint32_t *)d_codes);
    else if (l == 192) hipLaunchKernelGGL(bign_pubkey_val_kernel<12>, g, t, 0, st, (const uint8_t *)d_pubkeys, n, (uint32_t *)d_codes);
    else if (l == 256) hipLaunchKernelGGL(bign_pubkey_val_kernel<16>, g, t, 0, st, (const uint8_t *)d_pubkeys, n, (uint32_t *)d_codes);
    else return ERR_BAD_PARAMS;
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

#ifdef BEE2HIP_EXPERIMENTS
err_t launch_bign_debug_fe(size_t l, int op, const void *a, const void *b, void *out, size_t n, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    const dim3 g((unsigned)((n + 63) / 64)), t(64);
    if (l == 128) hipLaunchKernelGGL(bign_debug_fe_kernel<8>, g, t, 0, st, op, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, n);
    else if (l == 192) hipLaunchKernelGGL(bign_debug_fe_kernel<12>, g, t, 0, st, op, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, n);
    else if (l == 256) hipLaunchKernelGGL(bign_debug_fe_kernel<16>, g, t, 0, st, op, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, n);
    else return ERR_BAD_PARAMS;
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}
#endif

// this translation unit's copy of the belt S-box (belt_dev.hpp)
err_t upload_beltH_bign(const uint8_t *H)
{
    B2H_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_beltH), H, 256));
    return ERR_OK;
}

}  // namespace bee2hip
