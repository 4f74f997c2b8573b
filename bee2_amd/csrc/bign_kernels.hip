// bign_kernels.hip -- batched bign signature verification on bign-curve256v1 (gfx950).
//
// H3 of SURVEY.md 8a.  Replaces n calls of bign128Verify / bignVerifyEc
// (src/crypto/bign/bign128.c:177-185, src/crypto/bign/bign_sign.c:268-347).
// One lane per signature; per-signature result is the bee2 err_t the reference returns.
//
// The reference computes R = s1' G + (s0 + 2^128) Q with interleaved width-5 NAF
// (ecAddMulA, src/math/ec.c:1183-1273): data-dependent branching, 257 doublings.  Any
// correct algorithm yields the same affine R, so the GPU uses a wavefront-friendly
// schedule instead:
//   * G part: fixed-base comb, 16 windows x 16 bits, table of 16 x 65535 affine points
//     (64 MiB, built once per device: bign_gtable_kernel makes the 8-bit seed table by
//     double-and-add, bign_gtable16_kernel combines it) -> 16 mixed additions, NO
//     doublings for the 256-bit scalar;
//   * Q part: signed radix-16 digits of the 129-bit scalar (uniform 4 doublings + 1
//     addition per digit, 33 digits), per-signature table 1Q..8Q kept in an HBM scratch
//     laid out [entry][limb][signature] so table reads coalesce across the wavefront.
// Exceptional cases of the addition law (operand O, P = +-Q) cannot occur for honest
// inputs; lanes that hit one are flagged and recomputed by bign_slow_kernel with the
// complete (branchy) formulas, so verdicts are exact for every input.
//
// Kernels per batch (same stream): prep -> main -> slow -> tail.
//   prep : range checks (bign_sign.c:306-318), u = s1 + H mod q (:320-327),
//          v = s0 + 2^128 (:329-330), Q table
//   main : the double-scalar multiplication and x_R = X / Z^2 (one Fermat inversion)
//   slow : flagged lanes only
//   tail : belt-hash(oid || x_R || H) == s0 ? (bign_sign.c:337-343)
// HBM traffic is irrelevant here (148 B of input per ~7.5e5 VALU ops): the bound is the
// integer multiplier rate.
#include "belt_dev.hpp"
#include "bign_dev.hpp"
#include <mutex>
#include "common.hpp"

namespace bee2hip {

// status word per signature while the batch is in flight
constexpr uint32_t ST_PENDING = 0xFFFFFFFFu;     // fast path result in rx[]
constexpr uint32_t ST_SLOW = 0xFFFFFFFEu;        // needs the complete slow path
// anything else: a final bee2 err_t

// q (group order) and yG, STB 34.101.45 annex B.1 (bign_params.c:58-73), LE limbs
__constant__ uint32_t c_bign_q[8] = {0x263D6607u, 0x7E5ABF99u, 0x0DFB4DFCu, 0xD95C8ED6u,
                                     0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
__constant__ uint32_t c_bign_yG[8] = {0x04516A93u, 0x1E29CF18u, 0xC408F652u, 0x78913966u,
                                      0x51D6835Du, 0x5CE4C9A3u, 0xFB16D69Fu, 0x6BF7FC3Cu};

constexpr int GT_WINDOWS = 32;
constexpr int GT_ENTRIES = 256;                  // entry 0 unused (the neutral element)
// second-level comb: 16 windows x 16 bits, entry (w, b) = b * 2^(16 w) * G, 64 MiB of HBM
// (fits the 256 MiB Infinity Cache), built from the 8-bit table by one addition per entry
constexpr int GT16_WINDOWS = 16;
constexpr int GT16_ENTRIES = 65536;

struct VerifyScratch {          // all arrays are [..][n_pad] (signature index fastest)
    uint32_t *status;           // [n_pad]
    uint32_t *u;                // [8][n_pad]   scalar of G
    uint32_t *w;                // [5][n_pad]   v + 0x888..8 (33 nibbles): digit_i = nib_i - 8
    uint32_t *qtab;             // [8][24][n_pad] Jacobian 1Q..8Q
    uint32_t *rx;               // [8][n_pad]   canonical x_R
    size_t n_pad;
};

// ------------------------------------------------------------------ loaders ---
__device__ __forceinline__ void load_fe_bytes(fe &r, const uint8_t *p)
{
    // 32 little-endian octets -> 8 limbs (wwFrom, src/math/ww.c); p is 16-byte aligned
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    const uint4 b = *reinterpret_cast<const uint4 *>(p + 16);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
}
__device__ __forceinline__ void store_soa(uint32_t *base, size_t n_pad, size_t idx, const fe &a)
{
#pragma unroll
    for (int l = 0; l < 8; ++l) base[(size_t)l * n_pad + idx] = a.v[l];
}
__device__ __forceinline__ void load_soa(fe &a, const uint32_t *base, size_t n_pad, size_t idx)
{
#pragma unroll
    for (int l = 0; l < 8; ++l) a.v[l] = base[(size_t)l * n_pad + idx];
}
__device__ __forceinline__ void store_jac(const VerifyScratch &S, int e, size_t idx, const jac &P)
{
    uint32_t *b = S.qtab + (size_t)e * 24 * S.n_pad;
    store_soa(b, S.n_pad, idx, P.X);
    store_soa(b + 8 * S.n_pad, S.n_pad, idx, P.Y);
    store_soa(b + 16 * S.n_pad, S.n_pad, idx, P.Z);
}
__device__ __forceinline__ void load_jac(jac &P, const VerifyScratch &S, int e, size_t idx)
{
    const uint32_t *b = S.qtab + (size_t)e * 24 * S.n_pad;
    load_soa(P.X, b, S.n_pad, idx);
    load_soa(P.Y, b + 8 * S.n_pad, S.n_pad, idx);
    load_soa(P.Z, b + 16 * S.n_pad, S.n_pad, idx);
}

__device__ __forceinline__ void load_aff(aff &E, const uint4 *e)
{
    const uint4 x0 = e[0], x1 = e[1], y0 = e[2], y1 = e[3];
    E.x.v[0] = x0.x; E.x.v[1] = x0.y; E.x.v[2] = x0.z; E.x.v[3] = x0.w;
    E.x.v[4] = x1.x; E.x.v[5] = x1.y; E.x.v[6] = x1.z; E.x.v[7] = x1.w;
    E.y.v[0] = y0.x; E.y.v[1] = y0.y; E.y.v[2] = y0.z; E.y.v[3] = y0.w;
    E.y.v[4] = y1.x; E.y.v[5] = y1.y; E.y.v[6] = y1.z; E.y.v[7] = y1.w;
}

// --------------------------------------------------------------------- prep ---
__global__ __launch_bounds__(256)
void bign_prep_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                      const uint8_t *__restrict__ pubkeys, size_t n, VerifyScratch S)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;

    aff Q;
    load_fe_bytes(Q.x, pubkeys + 64 * idx);
    load_fe_bytes(Q.y, pubkeys + 64 * idx + 32);
    fe s1, H;
    load_fe_bytes(s1, sigs + 48 * idx + 16);
    load_fe_bytes(H, hashes + 32 * idx);
    const uint4 s0 = *reinterpret_cast<const uint4 *>(sigs + 48 * idx);

    uint32_t q[8], P[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { q[i] = c_bign_q[i]; P[i] = 0xFFFFFFFFu; }
    P[0] = P_LIMB0;

    // qrFrom rejects coordinates >= p (bign_sign.c:306-311); there is no on-curve check
    if (u256_ge(Q.x.v, P) || u256_ge(Q.y.v, P)) { S.status[idx] = ERR_BAD_PUBKEY; return; }
    // s1 >= q (bign_sign.c:313-318)
    if (u256_ge(s1.v, q)) { S.status[idx] = ERR_BAD_SIG; return; }

    // H <- H - q if H >= q ; u <- (s1 + H) mod q   (bign_sign.c:320-327)
    {
        uint32_t t[8];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t d = (uint64_t)H.v[i] - q[i] - borrow;
            t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) H.v[i] = borrow ? H.v[i] : t[i];
        uint64_t c = 0;
        uint32_t s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { c += (uint64_t)s1.v[i] + H.v[i]; s[i] = (uint32_t)c; c >>= 32; }
        const uint32_t carry = (uint32_t)c;
        borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t d = (uint64_t)s[i] - q[i] - borrow;
            t[i] = (uint32_t)d; borrow = (uint32_t)(d >> 32) & 1u;
        }
        const bool ge = carry || !borrow;            // s1 + H >= q
        fe u;
#pragma unroll
        for (int i = 0; i < 8; ++i) u.v[i] = ge ? t[i] : s[i];
        store_soa(S.u, S.n_pad, idx, u);
    }
    // v = s0 + 2^128 ; w = v + 0x8888...8 (33 nibbles) so that digit_i = nibble_i(w) - 8
    {
        uint64_t c = 0;
        const uint32_t v[5] = {s0.x, s0.y, s0.z, s0.w, 1u};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            c += (uint64_t)v[i] + (i < 4 ? 0x88888888u : 0x8u);
            S.w[(size_t)i * S.n_pad + idx] = (uint32_t)c;
            c >>= 32;
        }
    }
    // table 1Q..8Q (Jacobian).  Any exceptional case -> slow path.
    bool ok = true;
    jac P1, P2, P3, P4, T;
    P1.X = Q.x; P1.Y = Q.y; fe_set_one(P1.Z);
    store_jac(S, 0, idx, P1);
    P2 = P1; jac_dbl(P2);                 store_jac(S, 1, idx, P2);
    P3 = P2; ok &= jac_madd(P3, Q);       store_jac(S, 2, idx, P3);
    P4 = P2; jac_dbl(P4);                 store_jac(S, 3, idx, P4);
    T = P4;  ok &= jac_madd(T, Q);        store_jac(S, 4, idx, T);      // 5Q
    T = P3;  jac_dbl(T);                  store_jac(S, 5, idx, T);      // 6Q
    ok &= !fe_is_zero(T.Z);
    ok &= jac_madd(T, Q);                 store_jac(S, 6, idx, T);      // 7Q
    ok &= !fe_is_zero(T.Z);
    T = P4;  jac_dbl(T);                  store_jac(S, 7, idx, T);      // 8Q
    ok &= !fe_is_zero(T.Z) && !fe_is_zero(P2.Z) && !fe_is_zero(P3.Z) && !fe_is_zero(P4.Z);
    S.status[idx] = ok ? ST_PENDING : ST_SLOW;
}

// --------------------------------------------------------------------- main ---
__global__ __launch_bounds__(256)
void bign_main_kernel(size_t n, VerifyScratch S, const uint4 *__restrict__ gtab)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (S.status[idx] != ST_PENDING) return;

    uint32_t w[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) w[i] = S.w[(size_t)i * S.n_pad + idx];
    bool ok = true;

    // top digit d_32 = nibble_32(w) - 8 is 1 or 2
    jac T;
    load_jac(T, S, (int)(w[4] & 15u) - 9, idx);

#pragma unroll 1
    for (int i = 31; i >= 0; --i) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) jac_dbl(T);
        const int d = (int)(w[3] >> 28) - 8;            // next digit, in [-8, 7]
        w[3] = (w[3] << 4) | (w[2] >> 28);
        w[2] = (w[2] << 4) | (w[1] >> 28);
        w[1] = (w[1] << 4) | (w[0] >> 28);
        w[0] <<= 4;
        if (d != 0) {
            jac E;
            load_jac(E, S, (d < 0 ? -d : d) - 1, idx);
            if (d < 0) fe_neg(E.Y, E.Y);
            ok &= jac_add(T, E);
        }
    }
    // + u G : comb over the 16 halfwords of u (16 mixed additions, no doublings)
    fe u;
    load_soa(u, S.u, S.n_pad, idx);
#pragma unroll 1
    for (int win = 0; win < GT16_WINDOWS; ++win) {
        const uint32_t b = u.v[0] & 0xFFFFu;
#pragma unroll
        for (int l = 0; l < 7; ++l) u.v[l] = (u.v[l] >> 16) | (u.v[l + 1] << 16);
        u.v[7] >>= 16;
        if (b != 0) {
            aff E;
            load_aff(E, gtab + ((size_t)win * GT16_ENTRIES + b) * 4);
            ok &= jac_madd(T, E);
        }
    }
    ok &= !fe_is_zero(T.Z);
    if (!ok) { S.status[idx] = ST_SLOW; return; }
    // x_R = X / Z^2  (ecpToAJ, ecp_j.c:104-133)
    fe zi = fe_inv(T.Z);
    fe_sqr(zi, zi);
    fe_mul(zi, T.X, zi);
    fe_canon(zi, zi);
    store_soa(S.rx, S.n_pad, idx, zi);
}

// --------------------------------------------------------------------- slow ---
// complete, branchy double-and-add for flagged lanes: R = u G + v Q, every exceptional
// case of ecpAddJ / ecpDblJA3 handled.  Rare (never for honest inputs).
__device__ __forceinline__ bool bit_at(const uint32_t *k, int i) { return (k[i >> 5] >> (i & 31)) & 1u; }

__global__ __launch_bounds__(64)
void bign_slow_kernel(const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ pubkeys,
                      size_t n, VerifyScratch S)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (S.status[idx] != ST_SLOW) return;

    uint32_t u[8], v[5];
    for (int i = 0; i < 8; ++i) u[i] = S.u[(size_t)i * S.n_pad + idx];
    const uint4 s0 = *reinterpret_cast<const uint4 *>(sigs + 48 * idx);
    v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w; v[4] = 1u;

    jac G, Q, T;
    fe_set_zero(G.X);
    for (int i = 0; i < 8; ++i) G.Y.v[i] = c_bign_yG[i];
    fe_set_one(G.Z);
    load_fe_bytes(Q.X, pubkeys + 64 * idx);
    load_fe_bytes(Q.Y, pubkeys + 64 * idx + 32);
    fe_set_one(Q.Z);
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
#pragma unroll 1
    for (int i = 255; i >= 0; --i) {
        jac_dbl(T);
        if (bit_at(u, i)) jac_add_complete(T, G);
        if (i <= 128 && bit_at(v, i)) jac_add_complete(T, Q);
    }
    if (fe_is_zero(T.Z)) { S.status[idx] = ERR_BAD_SIG; return; }      // R == O (bign_sign.c:332-336)
    fe zi = fe_inv(T.Z);
    fe_sqr(zi, zi);
    fe_mul(zi, T.X, zi);
    fe_canon(zi, zi);
    store_soa(S.rx, S.n_pad, idx, zi);
    S.status[idx] = ST_PENDING;
}

// --------------------------------------------------------------------- tail ---
constexpr int OID_MAX = 128;                      // longest DER OID the kernel stages
struct OidArg { uint32_t len; uint8_t der[OID_MAX]; };
constexpr int TAIL_MSG_STRIDE = OID_MAX + 64 + 32;          // per-lane message area, zero padded

__global__ __launch_bounds__(64)
void bign_tail_kernel(const uint8_t *__restrict__ hashes, const uint8_t *__restrict__ sigs,
                      size_t n, VerifyScratch S, OidArg oid, uint32_t *__restrict__ codes)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tab[BeltTabSmall::kBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_msg[64 * TAIL_MSG_STRIDE];
    BeltTabSmall::fill(s_tab, threadIdx.x, 64);
    __syncthreads();
    const BeltTabSmall T(s_tab);

    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t st = S.status[idx];
    if (st != ST_PENDING) { codes[idx] = st; return; }

    // message = oid_der || <x_R>_256 || H  (bign_sign.c:339-342), staged per lane in LDS
    uint8_t *m = s_msg + threadIdx.x * TAIL_MSG_STRIDE;
    const uint32_t L = oid.len + 64;
    for (uint32_t i = 0; i < TAIL_MSG_STRIDE; i += 4) *reinterpret_cast<uint32_t *>(m + i) = 0;
    for (uint32_t i = 0; i < oid.len; ++i) m[i] = oid.der[i];
    for (int l = 0; l < 8; ++l) {
        const uint32_t x = S.rx[(size_t)l * S.n_pad + idx];
        for (int b = 0; b < 4; ++b) m[oid.len + 4 * l + b] = (uint8_t)(x >> (8 * b));
    }
    for (int i = 0; i < 32; ++i) m[oid.len + 32 + i] = hashes[32 * idx + i];

    // belt-hash (src/crypto/belt/belt_hash.c:43-171)
    uint32_t h[8], s[4] = {0, 0, 0, 0}, X[8], s1[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        h[i] = (uint32_t)c_beltH[4 * i] | (uint32_t)c_beltH[4 * i + 1] << 8 |
               (uint32_t)c_beltH[4 * i + 2] << 16 | (uint32_t)c_beltH[4 * i + 3] << 24;
    const uint32_t nblk = (L + 31) / 32;
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; ++b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) X[i] = *reinterpret_cast<const uint32_t *>(m + 32 * b + 4 * i);
        belt_compress(T, s1, h, X);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] ^= s1[i];
    }
    // final block: <bit length>_128 || s  (belt_hash.c:120-135, belt_lcl.c:25-51)
    X[0] = L << 3; X[1] = 0; X[2] = 0; X[3] = 0;
    X[4] = s[0]; X[5] = s[1]; X[6] = s[2]; X[7] = s[3];
    belt_compress(T, s1, h, X);
    const uint4 s0 = *reinterpret_cast<const uint4 *>(sigs + 48 * idx);
    const bool match = h[0] == s0.x && h[1] == s0.y && h[2] == s0.z && h[3] == s0.w;
    codes[idx] = match ? ERR_OK : ERR_BAD_SIG;
}

// ------------------------------------------------------------------- G table ---
// entry (win, b) = b * 2^(8 win) * G in affine form, b = 1..255.  One thread per entry,
// complete double-and-add; runs once per device.
__global__ __launch_bounds__(64)
void bign_gtable_kernel(uint4 *__restrict__ gtab)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= GT_WINDOWS * GT_ENTRIES) return;
    const int win = id / GT_ENTRIES, b = id % GT_ENTRIES;
    uint4 *e = gtab + (size_t)id * 4;
    if (b == 0) { e[0] = e[1] = e[2] = e[3] = make_uint4(0, 0, 0, 0); return; }
    jac G, T;
    fe_set_zero(G.X);
    for (int i = 0; i < 8; ++i) G.Y.v[i] = c_bign_yG[i];
    fe_set_one(G.Z);
    fe_set_zero(T.X); fe_set_one(T.Y); fe_set_zero(T.Z);
    const int top = 8 * win + 7;
#pragma unroll 1
    for (int i = top; i >= 0; --i) {
        jac_dbl(T);
        const int rel = i - 8 * win;
        if (rel >= 0 && ((b >> rel) & 1)) jac_add_complete(T, G);
    }
    fe zi = fe_inv(T.Z), zi2, x, y;
    fe_sqr(zi2, zi);
    fe_mul(x, T.X, zi2);
    fe_mul(zi2, zi2, zi);
    fe_mul(y, T.Y, zi2);
    fe_canon(x, x);
    fe_canon(y, y);
    e[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    e[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    e[2] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
    e[3] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
}

// 16-bit table from the 8-bit one: entry16(w, b) = entry8(2w, b & 255) + entry8(2w+1, b >> 8).
__global__ __launch_bounds__(256)
void bign_gtable16_kernel(const uint4 *__restrict__ gtab8, uint4 *__restrict__ gtab16)
{
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)GT16_WINDOWS * GT16_ENTRIES) return;
    const int win = (int)(id / GT16_ENTRIES), b = (int)(id % GT16_ENTRIES);
    const int lo = b & 255, hi = b >> 8;
    uint4 *e = gtab16 + id * 4;
    if (b == 0) { e[0] = e[1] = e[2] = e[3] = make_uint4(0, 0, 0, 0); return; }
    const uint4 *elo = gtab8 + ((size_t)(2 * win) * GT_ENTRIES + lo) * 4;
    const uint4 *ehi = gtab8 + ((size_t)(2 * win + 1) * GT_ENTRIES + hi) * 4;
    if (hi == 0 || lo == 0) {                     // one summand is the neutral element: copy
        const uint4 *src = hi == 0 ? elo : ehi;
        e[0] = src[0]; e[1] = src[1]; e[2] = src[2]; e[3] = src[3];
        return;
    }
    aff A, B;
    load_aff(A, elo);
    load_aff(B, ehi);
    jac T, E;
    T.X = A.x; T.Y = A.y; fe_set_one(T.Z);
    E.X = B.x; E.Y = B.y; fe_set_one(E.Z);
    jac_add_complete(T, E);                       // distinct multiples of G below the group order: never O
    fe zi = fe_inv(T.Z), zi2, x, y;
    fe_sqr(zi2, zi);
    fe_mul(x, T.X, zi2);
    fe_mul(zi2, zi2, zi);
    fe_mul(y, T.Y, zi2);
    fe_canon(x, x);
    fe_canon(y, y);
    e[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    e[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    e[2] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
    e[3] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
}

// --------------------------------------------------------- debug / self-test ---
// element-wise field ops over arrays of 8-limb values, used by tests/test_gpu_field.py
// to check the GF(p) layer against Python big integers.  op: 0 mul, 1 sqr, 2 add, 3 sub,
// 4 inv, 5 mul<3>, 6 sqr<8>, 7 canon, 8 dbl-point-x (a = X, b = Y, Z = 1 -> affine x of 2P)
__global__ void bign_debug_fe_kernel(int op, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    fe x, y, r;
    for (int i = 0; i < 8; ++i) { x.v[i] = a[8 * idx + i]; y.v[i] = b[8 * idx + i]; }
    switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sqr(r, x); break;
    case 2: fe_add(r, x, y); break;
    case 3: fe_sub(r, x, y); break;
    case 4: r = fe_inv(x); break;
    case 5: fe_mul<3>(r, x, y); break;
    case 6: fe_sqr<8>(r, x); break;
    case 7: r = x; break;
    default: {
        jac P; P.X = x; P.Y = y; fe_set_one(P.Z);
        jac_dbl(P);
        fe zi = fe_inv(P.Z);
        fe_sqr(zi, zi);
        fe_mul(r, P.X, zi);
    } break;
    }
    fe_canon(r, r);
    for (int i = 0; i < 8; ++i) out[8 * idx + i] = r.v[i];
}

// ------------------------------------------------------------------ host side ---
struct BignDevice {
    uint4 *gtab = nullptr;            // 64 MiB 16-bit comb table (the 512 KiB 8-bit one is its seed)
};
static BignDevice g_bign[64];
static std::mutex g_bign_mu;          // table / scratch bookkeeping is per device, shared by threads

static err_t bign_device(BignDevice **out, hipStream_t st)
{
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return ERR_BAD_INPUT;
    BignDevice &D = g_bign[dev];
    if (!D.gtab) {
        uint4 *t8 = nullptr, *t16 = nullptr;
        if (hipMalloc((void **)&t8, (size_t)GT_WINDOWS * GT_ENTRIES * 64) != hipSuccess) return ERR_OUTOFMEMORY;
        if (hipMalloc((void **)&t16, (size_t)GT16_WINDOWS * GT16_ENTRIES * 64) != hipSuccess) {
            (void)hipFree(t8);
            return ERR_OUTOFMEMORY;
        }
        hipLaunchKernelGGL(bign_gtable_kernel, dim3(GT_WINDOWS * GT_ENTRIES / 64), dim3(64), 0, st, t8);
        hipLaunchKernelGGL(bign_gtable16_kernel, dim3(GT16_WINDOWS * GT16_ENTRIES / 256), dim3(256), 0, st,
                           (const uint4 *)t8, t16);
        B2H_TRY(hipGetLastError());
        B2H_TRY(hipStreamSynchronize(st));
        (void)hipFree(t8);
        D.gtab = t16;
    }
    *out = &D;
    return ERR_OK;
}

static err_t bign_scratch(hipStream_t st, size_t n, VerifyScratch &S)
{
    const size_t n_pad = (n + 63) & ~(size_t)63;
    const size_t words = n_pad * (1 + 8 + 5 + 8 * 24 + 8);
    void *base = nullptr;
    err_t code = scratch_for_stream(st, 0, words * 4, &base);
    if (code != ERR_OK) return code;
    uint32_t *p = (uint32_t *)base;
    S.n_pad = n_pad;
    S.status = p; p += n_pad;
    S.u = p; p += 8 * n_pad;
    S.w = p; p += 5 * n_pad;
    S.qtab = p; p += 8 * 24 * n_pad;
    S.rx = p;
    return ERR_OK;
}

err_t launch_bign_verify(const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                         const void *d_sigs, const void *d_pubkeys, size_t n, void *d_codes,
                         hipStream_t st)
{
    if (n == 0) return ERR_OK;
    if (oid_len > OID_MAX) return ERR_NOT_IMPLEMENTED;
    BignDevice *D = nullptr;
    err_t code;
    {
        std::lock_guard<std::mutex> lk(g_bign_mu);     // table construction happens once per device
        code = bign_device(&D, st);
    }
    if (code != ERR_OK) return code;
    VerifyScratch S;
    code = bign_scratch(st, n, S);
    if (code != ERR_OK) return code;
    OidArg oid;
    memset(&oid, 0, sizeof oid);
    oid.len = (uint32_t)oid_len;
    memcpy(oid.der, oid_der, oid_len);
    const unsigned g256 = (unsigned)((n + 255) / 256), g64 = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(bign_prep_kernel, dim3(g256), dim3(256), 0, st, (const uint8_t *)d_hashes,
                       (const uint8_t *)d_sigs, (const uint8_t *)d_pubkeys, n, S);
    hipLaunchKernelGGL(bign_main_kernel, dim3(g256), dim3(256), 0, st, n, S, (const uint4 *)D->gtab);
    hipLaunchKernelGGL(bign_slow_kernel, dim3(g64), dim3(64), 0, st, (const uint8_t *)d_sigs,
                       (const uint8_t *)d_pubkeys, n, S);
    hipLaunchKernelGGL(bign_tail_kernel, dim3(g64), dim3(64), 0, st, (const uint8_t *)d_hashes,
                       (const uint8_t *)d_sigs, n, S, oid, (uint32_t *)d_codes);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

err_t launch_bign_debug_fe(int op, const void *a, const void *b, void *out, size_t n, hipStream_t st)
{
    if (n == 0) return ERR_OK;
    hipLaunchKernelGGL(bign_debug_fe_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, op,
                       (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, n);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
}

}  // namespace bee2hip
