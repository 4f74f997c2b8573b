// host_small.hpp -- the drop-in layer's HOST path for SMALL single calls (product code; plain C++17, no HIP).
//
// Why it exists ("symbols a C-ABI replacement must export ... single-call path; may run on CPU"): a bee2
// caller relinked against libbee2hip.so calls bashF on ONE state, beltCTRStepE on 16 bytes, beltHashStepH on a
// 32 KiB chunk.  One such call through the GPU costs a launch and a synchronise (19-35 us; a serial chain such as
// one message's sponge or CBC-MAC runs on a single lane at 4-6 MB/s) where one host core needs 0.3-0.5 us per
// primitive: VERDICT r02 "missing 3".  What is here is used ONLY
//   * by the bee2 drop-in symbols (capi.hip), never by a bee2hip_*_batch / *_dev / *_multi entry point, bench.py's
//     timed regions or anything that takes a batch;
//   * below the crossover of the call at hand (capi.hip host_wanted(): a single primitive, parallel modes under
//     8 KiB per call, serial chains of one message at every size), unless BEE2HIP_FORCE=gpu|cpu says otherwise;
//   * after the calling thread has initialised its HIP device: a process without a GPU fails as loudly as before.
// It is an independent statement of STB 34.101.77 (bash-f), 34.101.31 (belt) -- not the oracle (oracle/ is test
// infrastructure and is never linked here), not the reference's code: every function cites the lines of the
// reference whose behaviour it must reproduce, and tests/test_host_small.py pins each one to the oracle and the
// golden vectors on CPU, tests/test_gpu_*.py run every drop-in fixture through both paths on the GPU box.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#define BEE2HIP_HOST_X86_CLMUL 1      // gf_mul_clmul below: PCLMULQDQ, picked at run time (never in the device pass)
#include <emmintrin.h>
#include <wmmintrin.h>
#endif

namespace bee2hip {
namespace hostp {

// ------------------------------------------------------------------ bash-f ---
// src/crypto/bash/bash_f64.c:32-44 (S-box), :100-134 (round, word permutation), :50-84 (constants)
static inline uint64_t rotl64(uint64_t x, unsigned r) { r &= 63u; return r ? (x << r) | (x >> (64u - r)) : x; }

struct BashConsts {
    uint64_t c[24];
    unsigned m1[8], n1[8], m2[8], n2[8];
    constexpr BashConsts() : c(), m1(), n1(), m2(), n2()
    {
        uint64_t x = 0x3BF5080AC8BA94B1ull;
        for (int i = 0; i < 24; ++i) {
            c[i] = x;
            x = (x >> 1) ^ ((x & 1u) ? 0xDC2BE1997FE0D8AEull : 0ull);
        }
        unsigned a = 8, b = 53, d = 14, e = 1;
        for (int j = 0; j < 8; ++j) {
            m1[j] = a; n1[j] = b; m2[j] = d; n2[j] = e;
            a = a * 7u % 64u; b = b * 7u % 64u; d = d * 7u % 64u; e = e * 7u % 64u;
        }
    }
};

// one permutation of a 24-word state held as native integers
static inline void bash_f_words(uint64_t s[24])
{
    static constexpr BashConsts B{};
    static constexpr int P0[8] = {6, 3, 0, 5, 2, 7, 4, 1}, P1[8] = {7, 2, 1, 4, 3, 6, 5, 0}, P2[8] = {1, 0, 3, 2, 5, 4, 7, 6};
    for (int round = 0; round < 24; ++round) {
        uint64_t r0[8], r1[8], r2[8];
        for (int j = 0; j < 8; ++j) {
            const uint64_t w0 = s[j], w1 = s[8 + j], w2 = s[16 + j];
            const uint64_t u0 = w0 ^ w1 ^ w2;
            const uint64_t t = w1 ^ rotl64(u0, B.n1[j]);
            const uint64_t u1 = t ^ rotl64(w0, B.m1[j]);
            const uint64_t u2 = w2 ^ rotl64(w2, B.m2[j]) ^ rotl64(t, B.n2[j]);
            r0[j] = u0 ^ (~u2 | u1);
            r1[j] = u1 ^ (u0 | u2);
            r2[j] = u2 ^ (u0 & u1);
        }
        for (int i = 0; i < 8; ++i) {
            s[i] = r1[P1[i]];
            s[8 + i] = r2[P2[i]];
            s[16 + i] = r0[P0[i]];
        }
        s[23] ^= B.c[round];
    }
}
static inline uint64_t ld64le(const uint8_t *p)
{
    uint64_t v = 0;
    for (int k = 7; k >= 0; --k) v = (v << 8) | p[k];
    return v;
}
static inline void st64le(uint8_t *p, uint64_t v)
{
    for (int k = 0; k < 8; ++k) p[k] = (uint8_t)(v >> (8 * k));
}
// bashF on an octet string in bee2's layout (word k at bytes [8k, 8k + 8), little-endian; bash.h:133-136)
static inline void bashF(uint8_t block[192])
{
    uint64_t s[24];
    for (int k = 0; k < 24; ++k) s[k] = ld64le(block + 8 * k);
    bash_f_words(s);
    for (int k = 0; k < 24; ++k) st64le(block + 8 * k, s[k]);
}
// absorb `count` bytes into a sponge state: overwrite at pos, permute at buf_len (bashHashStepH, bash_hash.c:52-79)
static inline void sponge_absorb(uint8_t s[192], size_t buf_len, size_t *pos, const uint8_t *buf, size_t count)
{
    size_t p = *pos;
    while (count) {
        size_t take = buf_len - p;
        if (take > count) take = count;
        memcpy(s + p, buf, take);
        p += take; buf += take; count -= take;
        if (p == buf_len) { bashF(s); p = 0; }
    }
    *pos = p;
}

// -------------------------------------------------------------------- belt ---
// belt_block.c:121-269 (G-boxes, round, E), :286-295 (D).  T[r][b] = rotl32(H[b], 5 + 8 r).
struct BeltTables { uint32_t t[4][256]; };
static inline uint32_t rotl32(uint32_t x, unsigned r) { return (x << r) | (x >> (32u - r)); }
static inline void belt_tables(BeltTables &T, const uint8_t H[256])
{
    for (int r = 0; r < 4; ++r)
        for (int b = 0; b < 256; ++b) T.t[r][b] = rotl32((uint32_t)H[b], 5u + 8u * (unsigned)r);
}
// G_{5 + 8 R}(x): byte k of x goes through the table of rotation 5 + 8 ((R + k) mod 4)
template <int R>
static inline uint32_t belt_g(const BeltTables &T, uint32_t x)
{
    return T.t[R & 3][x & 255u] ^ T.t[(R + 1) & 3][(x >> 8) & 255u] ^ T.t[(R + 2) & 3][(x >> 16) & 255u] ^
           T.t[(R + 3) & 3][x >> 24];
}
// steps 2.1-2.9 of a round with the seven subkeys k[0..6] (belt_block.c:231-240)
static inline void belt_round(const BeltTables &T, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, const uint32_t k[7],
                              uint32_t i)
{
    b ^= belt_g<0>(T, a + k[0]);
    c ^= belt_g<2>(T, d + k[1]);
    a -= belt_g<1>(T, b + k[2]);
    const uint32_t e = belt_g<2>(T, b + c + k[3]) ^ i;
    b += e;
    c -= e;
    d += belt_g<1>(T, c + k[4]);
    b ^= belt_g<2>(T, a + k[5]);
    c ^= belt_g<0>(T, d + k[6]);
}
static inline void belt_encr(const BeltTables &T, uint32_t x[4], const uint32_t K[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    for (uint32_t i = 1; i <= 8; ++i) {
        uint32_t k[7];
        for (uint32_t j = 0; j < 7; ++j) k[j] = K[(7 * i - 7 + j) & 7];
        belt_round(T, a, b, c, d, k, i);
        const uint32_t ta = a, tb = b, tc = c, td = d;              // (a, b, c, d) <- (b, d, a, c)
        a = tb; b = td; c = ta; d = tc;
    }
    x[0] = b; x[1] = d; x[2] = a; x[3] = c;                          // belt_block.c:267-269
}
static inline void belt_decr(const BeltTables &T, uint32_t x[4], const uint32_t K[8])
{
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    for (uint32_t i = 8; i >= 1; --i) {
        uint32_t k[7];
        for (uint32_t j = 0; j < 7; ++j) k[j] = K[(7 * i - 1 - j) & 7];   // subkeys of round i in reverse order
        belt_round(T, a, b, c, d, k, i);
        const uint32_t ta = a, tb = b, tc = c, td = d;              // (a, b, c, d) <- (c, a, d, b)
        a = tc; b = ta; c = td; d = tb;
    }
    x[0] = c; x[1] = a; x[2] = d; x[3] = b;                          // belt_block.c:293-295
}
static inline uint32_t ld32le(const uint8_t *p)
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
static inline void st32le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static inline void ld_block(uint32_t x[4], const uint8_t *p) { for (int i = 0; i < 4; ++i) x[i] = ld32le(p + 4 * i); }
static inline void st_block(uint8_t *p, const uint32_t x[4]) { for (int i = 0; i < 4; ++i) st32le(p + 4 * i, x[i]); }

// (c + add) mod 2^128 on four little-endian words (beltBlockIncU32 applied `add` times, belt_ctr.c:27-35)
static inline void ctr_add(uint32_t c[4], uint64_t add)
{
    const uint64_t lo = (uint64_t)c[0] | (uint64_t)c[1] << 32;
    uint64_t hi = (uint64_t)c[2] | (uint64_t)c[3] << 32;
    const uint64_t nlo = lo + add;
    hi += nlo < lo;
    c[0] = (uint32_t)nlo; c[1] = (uint32_t)(nlo >> 32); c[2] = (uint32_t)hi; c[3] = (uint32_t)(hi >> 32);
}
// the block loop of beltCTRStepE (belt_ctr.c:85-110) on `count` bytes from a block boundary of the gamma: whole blocks,
// then a partial one whose unused gamma stays in block[] (reserved = 16 - tail).  ctr / block / reserved as bee2 leaves them.
static inline void ctr_blocks(const BeltTables &T, uint8_t *buf, size_t count, const uint32_t key[8], uint32_t ctr[4],
                              uint8_t block[16], size_t *reserved)
{
    while (count) {
        ctr_add(ctr, 1);
        uint32_t g[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
        belt_encr(T, g, key);
        st_block(block, g);
        const size_t take = count < 16 ? count : 16;
        for (size_t i = 0; i < take; ++i) buf[i] ^= block[i];
        buf += take; count -= take;
        *reserved = 16 - take;
    }
}

// belt-mac (belt_mac.c:58-138): modes as belt_mac_kernel (mixed_kernels.hip) -- 1 start, 2 absorb, 4 finalise into mac[]
static inline void mac_step(const BeltTables &T, const uint32_t key[8], uint32_t s[4], uint32_t r[4], uint32_t mac[4],
                            uint8_t block[16], size_t *filled, const uint8_t *buf, size_t count, int mode)
{
    if (mode & 1) {
        for (int k = 0; k < 4; ++k) { s[k] = 0; r[k] = 0; }
        belt_encr(T, r, key);
        *filled = 0;
    }
    if (mode & 2) {
        size_t f = *filled;
        if (f == 16 && count) {                  // whole blocks straight from the caller's buffer: the buffered block goes first, and
            uint32_t b[4];                       // the LAST 1..16 octets of buf stay behind in `block` (belt_mac.c:75-99)
            ld_block(b, block);
            for (int q = 0; q < 4; ++q) s[q] ^= b[q];
            belt_encr(T, s, key);
            f = 0;
        }
        if (f == 0)
            while (count > 16) {
                uint32_t b[4];
                ld_block(b, buf);
                for (int q = 0; q < 4; ++q) s[q] ^= b[q];
                belt_encr(T, s, key);
                buf += 16; count -= 16;
            }
        for (size_t k = 0; k < count; ++k) {
            if (f == 16) {                       // the buffered block is absorbed only when more data follows
                uint32_t b[4];
                ld_block(b, block);
                for (int q = 0; q < 4; ++q) s[q] ^= b[q];
                belt_encr(T, s, key);
                f = 0;
            }
            block[f++] = buf[k];
        }
        *filled = f;
    }
    if (mode & 4) {
        uint32_t b[4], m[4];
        const size_t f = *filled;
        if (f == 16) {
            ld_block(b, block);
            for (int q = 0; q < 4; ++q) m[q] = s[q] ^ b[q];
            m[0] ^= r[1]; m[1] ^= r[2]; m[2] ^= r[3]; m[3] ^= r[0] ^ r[1];          // phi1, belt_mac.c:112-115
        } else {
            block[f] = 0x80;                                                         // padded in place, belt_mac.c:123-124
            for (size_t k = f + 1; k < 16; ++k) block[k] = 0;
            ld_block(b, block);
            for (int q = 0; q < 4; ++q) m[q] = s[q] ^ b[q];
            m[0] ^= r[0] ^ r[3]; m[1] ^= r[0]; m[2] ^= r[1]; m[3] ^= r[2];          // phi2, belt_mac.c:129-132
        }
        belt_encr(T, m, key);
        for (int q = 0; q < 4; ++q) mac[q] = m[q];
    }
}

// belt-compress (belt_compr.c:27-87): s1 = E_X(h0 ^ h1) ^ h0 ^ h1; h0' = E_{s1 || h1}(X0) ^ X0; h1' = E_{~s1 || h0}(X1) ^ X1
static inline void belt_compress(const BeltTables &T, uint32_t s1[4], uint32_t h[8], const uint32_t X[8])
{
    uint32_t u[4], k1[8], k2[8], y0[4], y1[4];
    for (int i = 0; i < 4; ++i) { u[i] = h[i] ^ h[4 + i]; s1[i] = u[i]; }
    belt_encr(T, s1, X);
    for (int i = 0; i < 4; ++i) {
        s1[i] ^= u[i];
        k1[i] = s1[i]; k1[4 + i] = h[4 + i];
        k2[i] = ~s1[i]; k2[4 + i] = h[i];
        y0[i] = X[i]; y1[i] = X[4 + i];
    }
    belt_encr(T, y0, k1);
    belt_encr(T, y1, k2);
    for (int i = 0; i < 4; ++i) { h[i] = y0[i] ^ X[i]; h[4 + i] = y1[i] ^ X[4 + i]; }
}
// hs = h[8] || s[4]; nblocks whole 32-byte blocks, then (fin) the block <bit length>_128 || s (belt_hash.c:74-135)
static inline void hash_stream(const BeltTables &T, uint32_t hs[12], const uint8_t *data, size_t nblocks, int fin,
                               uint64_t bits_lo, uint64_t bits_hi)
{
    uint32_t X[8], s1[4];
    for (size_t b = 0; b < nblocks; ++b) {
        for (int k = 0; k < 8; ++k) X[k] = ld32le(data + 32 * b + 4 * k);
        belt_compress(T, s1, hs, X);
        for (int k = 0; k < 4; ++k) hs[8 + k] ^= s1[k];
    }
    if (fin) {
        X[0] = (uint32_t)bits_lo; X[1] = (uint32_t)(bits_lo >> 32); X[2] = (uint32_t)bits_hi; X[3] = (uint32_t)(bits_hi >> 32);
        for (int k = 0; k < 4; ++k) X[4 + k] = hs[8 + k];
        belt_compress(T, s1, hs, X);
    }
}

// ---------------------------------------------------------------- GF(2^128) ---
// GF(2)[x] / (x^128 + x^7 + x^2 + x + 1), bit i of the little-endian block = coefficient of x^i
// (beltPolyMul / beltBlockMulC, belt_lcl.c:99-132)
struct Gf { uint64_t lo, hi; };
static inline Gf gf_from(const uint32_t w[4]) { return Gf{(uint64_t)w[0] | (uint64_t)w[1] << 32, (uint64_t)w[2] | (uint64_t)w[3] << 32}; }
static inline void gf_to(uint32_t w[4], Gf a)
{
    w[0] = (uint32_t)a.lo; w[1] = (uint32_t)(a.lo >> 32); w[2] = (uint32_t)a.hi; w[3] = (uint32_t)(a.hi >> 32);
}
static inline Gf gf_mulx(Gf a)
{
    const uint64_t out = a.hi >> 63;
    a.hi = (a.hi << 1) | (a.lo >> 63);
    a.lo = (a.lo << 1) ^ (out ? 0x87ull : 0ull);
    return a;
}
// a * b in GF(2^128) = GF(2)[x] / (x^128 + x^7 + x^2 + x + 1), bit i of the little-endian 128-bit value = coefficient of
// x^i (beltPolyMul, belt_lcl.c:119-132 -> ppMul + ppRedBelt).  Bit-serial: the plain definition, kept as the reference the
// two fast forms below are tested against (tests/test_host_small.py).
static inline Gf gf_mul_bitserial(Gf a, Gf b)
{
    Gf r{0, 0};
    for (int i = 0; i < 128; ++i) {
        const uint64_t bit = ((i < 64 ? b.lo >> i : b.hi >> (i - 64)) & 1ull);
        if (bit) { r.lo ^= a.lo; r.hi ^= a.hi; }
        a = gf_mulx(a);
    }
    return r;
}
// Multiplication by a FIXED r (belt-dwp / belt-che multiply every block by the same r): 16 multiples of r once, then 32
// steps "acc <- acc * x^4 + nibble * r" from the top nibble down; the four bits a step pushes out fold back through a
// 16-entry table of (bits * (x^7 + x^2 + x + 1)).  ~25 ns per block where the bit-serial form takes 400.
struct GfMulTab {
    Gf m[16];
    explicit GfMulTab(Gf r)
    {
        m[0] = Gf{0, 0};
        m[1] = r;
        for (int i = 2; i < 16; i += 2) {
            m[i] = gf_mulx(m[i >> 1]);
            m[i + 1] = Gf{m[i].lo ^ r.lo, m[i].hi ^ r.hi};
        }
    }
    Gf mul(Gf a) const
    {
        // out4 * 0x87 for out4 = 0 .. 15 (no carry between the bits: 0x87 << 3 = 0x438 still fits)
        static const uint16_t red[16] = {0x000, 0x087, 0x10E, 0x189, 0x21C, 0x29B, 0x312, 0x395,
                                         0x438, 0x4BF, 0x536, 0x5B1, 0x624, 0x6A3, 0x72A, 0x7AD};
        Gf acc{0, 0};
        for (int i = 31; i >= 0; --i) {
            const unsigned out = (unsigned)(acc.hi >> 60);
            acc.hi = (acc.hi << 4) | (acc.lo >> 60);
            acc.lo = (acc.lo << 4) ^ red[out];
            const unsigned nib = (unsigned)((i < 16 ? a.lo >> (4 * i) : a.hi >> (4 * (i - 16))) & 15u);
            acc.lo ^= m[nib].lo; acc.hi ^= m[nib].hi;
        }
        return acc;
    }
};
#ifdef BEE2HIP_HOST_X86_CLMUL
// the same product with the carry-less multiplier of the host CPU (PCLMULQDQ): four 64 x 64 products, then the upper half
// folded back twice through x^128 = x^7 + x^2 + x + 1.  Chosen at run time (gf_have_clmul); ~3 ns per block.
__attribute__((target("pclmul,sse2"))) static inline Gf gf_mul_clmul(Gf a, Gf b)
{
    const __m128i va = _mm_set_epi64x((long long)a.hi, (long long)a.lo), vb = _mm_set_epi64x((long long)b.hi, (long long)b.lo);
    const __m128i p = _mm_set_epi64x(0, 0x87);
    const __m128i t0 = _mm_clmulepi64_si128(va, vb, 0x00), t1 = _mm_clmulepi64_si128(va, vb, 0x11);
    const __m128i t2 = _mm_xor_si128(_mm_clmulepi64_si128(va, vb, 0x10), _mm_clmulepi64_si128(va, vb, 0x01));
    __m128i lo = _mm_xor_si128(t0, _mm_slli_si128(t2, 8)), hi = _mm_xor_si128(t1, _mm_srli_si128(t2, 8));
    const __m128i r0 = _mm_clmulepi64_si128(hi, p, 0x00), r1 = _mm_clmulepi64_si128(hi, p, 0x01);    // hi.lo * f', hi.hi * f' (at x^64)
    lo = _mm_xor_si128(lo, _mm_xor_si128(r0, _mm_slli_si128(r1, 8)));
    const __m128i over = _mm_srli_si128(r1, 8);                                                     // < 2^8: what passed x^128 again
    lo = _mm_xor_si128(lo, _mm_clmulepi64_si128(over, p, 0x00));
    Gf r;
    r.lo = (uint64_t)_mm_cvtsi128_si64(lo);
    r.hi = (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(lo, 8));
    return r;
}
static inline bool gf_have_clmul()
{
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse2");
    return have;
}
#else
static inline Gf gf_mul_clmul(Gf a, Gf b) { return gf_mul_bitserial(a, b); }
static inline bool gf_have_clmul() { return false; }
#endif
static inline Gf gf_mul(Gf a, Gf b) { return gf_have_clmul() ? gf_mul_clmul(a, b) : gf_mul_bitserial(a, b); }
// t <- (t ^ X) * r over the 16-byte blocks of data, the last one zero-padded (belt_dwp.c:96-101,118-119)
// form: 0 = best available (the product), 1 = table, 2 = bit-serial (tests compare all three)
static inline void polyhash(uint32_t t[4], const uint32_t r[4], const uint8_t *data, size_t nbytes, int form = 0)
{
    Gf acc = gf_from(t);
    const Gf rr = gf_from(r);
    const bool clmul = form == 0 && gf_have_clmul();
    const bool table = !clmul && form != 2 && nbytes > 64;        // 16 multiples cost ~15 doublings: pays from a few blocks on
    GfMulTab tab(table ? rr : Gf{0, 0});
    while (nbytes) {
        uint8_t blk[16] = {0};
        const size_t take = nbytes < 16 ? nbytes : 16;
        memcpy(blk, data, take);
        uint32_t w[4];
        ld_block(w, blk);
        const Gf x = gf_from(w);
        acc.lo ^= x.lo; acc.hi ^= x.hi;
        acc = clmul ? gf_mul_clmul(acc, rr) : table ? tab.mul(acc) : gf_mul_bitserial(acc, rr);
        data += take; nbytes -= take;
    }
    gf_to(t, acc);
}

// ------------------------------------------------------- block-parallel modes ---
// mode 0: ECB encrypt, 1: ECB decrypt (in place), 2: CBC decrypt with chaining value iv (belt_ecb.c:63-107, belt_cbc.c:101-116)
static inline void modes_blocks(const BeltTables &T, int mode, uint8_t *buf, size_t nblocks, const uint32_t key[8],
                                const uint32_t iv[4])
{
    uint32_t prev[4] = {iv[0], iv[1], iv[2], iv[3]};
    for (size_t b = 0; b < nblocks; ++b) {
        uint32_t x[4], c[4];
        ld_block(x, buf + 16 * b);
        for (int i = 0; i < 4; ++i) c[i] = x[i];
        if (mode == 0) belt_encr(T, x, key); else belt_decr(T, x, key);
        if (mode == 2) for (int i = 0; i < 4; ++i) { x[i] ^= prev[i]; prev[i] = c[i]; }
        st_block(buf + 16 * b, x);
    }
}
// CBC encryption of whole blocks; chain = previous ciphertext block in / last ciphertext block out (belt_cbc.c:75-84)
static inline void cbc_encr_blocks(const BeltTables &T, uint8_t *buf, size_t nblocks, const uint32_t key[8], uint8_t chain[16])
{
    uint32_t prev[4];
    ld_block(prev, chain);
    for (size_t b = 0; b < nblocks; ++b) {
        uint32_t x[4];
        ld_block(x, buf + 16 * b);
        for (int i = 0; i < 4; ++i) x[i] ^= prev[i];
        belt_encr(T, x, key);
        st_block(buf + 16 * b, x);
        for (int i = 0; i < 4; ++i) prev[i] = x[i];
    }
    st_block(chain, prev);
}
// belt-bde (belt_bde.c:50-90): per block s <- s * x, Y = E/D(X ^ s) ^ s; s advanced in place
static inline void bde_blocks(const BeltTables &T, int decr, uint8_t *buf, size_t nblocks, const uint32_t key[8], uint32_t s[4])
{
    Gf t = gf_from(s);
    for (size_t b = 0; b < nblocks; ++b) {
        t = gf_mulx(t);
        uint32_t tw[4], x[4];
        gf_to(tw, t);
        ld_block(x, buf + 16 * b);
        for (int i = 0; i < 4; ++i) x[i] ^= tw[i];
        if (decr) belt_decr(T, x, key); else belt_encr(T, x, key);
        for (int i = 0; i < 4; ++i) x[i] ^= tw[i];
        st_block(buf + 16 * b, x);
    }
    gf_to(s, t);
}
// belt-che keystream (belt_che.c:69-97): per block s <- s * x ^ 1, Y = X ^ E_K(s); s advanced in place
static inline void che_blocks(const BeltTables &T, uint8_t *buf, size_t nblocks, const uint32_t key[8], uint32_t s[4])
{
    Gf t = gf_from(s);
    for (size_t b = 0; b < nblocks; ++b) {
        t = gf_mulx(t);
        t.lo ^= 1ull;
        uint32_t g[4], x[4];
        gf_to(g, t);
        belt_encr(T, g, key);
        ld_block(x, buf + 16 * b);
        for (int i = 0; i < 4; ++i) x[i] ^= g[i];
        st_block(buf + 16 * b, x);
    }
    gf_to(s, t);
}
// belt-wbl on n >= 2 blocks (STB 34.101.31 6.2 / belt_wbl.c: 2n rounds, round i: s <- r_1 ^ ... ^ r_{n-1};
// r* <- r_n ^ E_K(s) ^ <i>_128; (r_1 .. r_n) <- (r_2 .. r_{n-1}, r*, s)); decryption runs the rounds backwards.
// Rolling form (the one of belt_sde_kernel, belt_kernels.hip): the blocks stay where they are and a cyclic head h names the
// position of the logical r_1; the block a round needs as r_n is the s the previous round wrote, so it is carried in registers,
// and the XOR of r_1 .. r_{n-1} is updated by the two blocks that change instead of recomputed -- one block read and one block
// written per round instead of n (round 4: bee2's own beltBench::belt-sde through the drop-in 0.91 -> above the reference).
// After 2n rounds h is back where it started.
static inline void wbl(const BeltTables &T, int decr, uint8_t *a, size_t n, const uint32_t key[8])
{
    const uint64_t rounds = 2ull * n;
    uint32_t cur[4] = {0, 0, 0, 0}, w[4], e[4], x[4];
    for (size_t q = 0; q + 1 < n; ++q) { ld_block(w, a + 16 * q); for (int k = 0; k < 4; ++k) cur[k] ^= w[k]; }
    if (!decr) {
        uint32_t prev[4];
        ld_block(prev, a + 16 * (n - 1));                       // logical r_n
        size_t h = 0;
        for (uint64_t i = 1; i <= rounds; ++i) {
            for (int k = 0; k < 4; ++k) e[k] = cur[k];
            belt_encr(T, e, key);
            e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32);
            for (int k = 0; k < 4; ++k) x[k] = prev[k] ^ e[k];  // r* = r_n ^ E_K(s) ^ <i>
            ld_block(w, a + 16 * h);                            // the r_1 that leaves the sum
            st_block(a + 16 * (h ? h - 1 : n - 1), x);
            for (int k = 0; k < 4; ++k) { prev[k] = cur[k]; cur[k] ^= w[k] ^ x[k]; }
            h = h + 1 == n ? 0 : h + 1;
        }
        st_block(a + 16 * (n - 1), prev);                       // h == 0 again
    } else {
        uint32_t sv[4];
        ld_block(sv, a + 16 * (n - 1));                         // s = logical r_n
        size_t h = n - 1;
        for (uint64_t i = rounds; i >= 1; --i) {
            for (int k = 0; k < 4; ++k) e[k] = sv[k];
            belt_encr(T, e, key);
            e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32);
            ld_block(x, a + 16 * (h ? h - 1 : n - 1));          // r*
            for (int k = 0; k < 4; ++k) w[k] = cur[k] ^ sv[k] ^ x[k];
            st_block(a + 16 * h, w);                            // the r_1 of before the round
            for (int k = 0; k < 4; ++k) { cur[k] = sv[k]; sv[k] = x[k] ^ e[k]; }
            h = h ? h - 1 : n - 1;
        }
        st_block(a + 16 * h, sv);                               // h == n - 1 again
    }
}
// belt-sde on one sector (belt_sde.c:47-76): XEX around belt-wbl with the tweak E_K(iv) on the first block
static inline void sde_sector(const BeltTables &T, int decr, uint8_t *buf, size_t count, const uint8_t iv[16], const uint32_t key[8])
{
    uint32_t tw[4], x[4];
    ld_block(tw, iv);
    belt_encr(T, tw, key);
    ld_block(x, buf);
    for (int k = 0; k < 4; ++k) x[k] ^= tw[k];
    st_block(buf, x);
    wbl(T, decr, buf, count / 16, key);
    ld_block(x, buf);
    for (int k = 0; k < 4; ++k) x[k] ^= tw[k];
    st_block(buf, x);
}

}  // namespace hostp
}  // namespace bee2hip
