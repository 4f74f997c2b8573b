/*
 * belt_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of STB 34.101.31 belt: S-box, block encryption, CTR, MAC,
 * compress and hash.  Written from the algorithm as pinned by the STB vectors;
 * one S-box byte table + explicit G-function (no pre-rotated tables), explicit
 * register rotation per round (no macro-argument permutation).
 *
 * Follows:
 *   H generator      src/crypto/belt/belt_block.c:21-35 (comment recipe), :43-60
 *   beltKeyExpand2   src/crypto/belt/belt_block.c:88-106
 *   G / R / E        src/crypto/belt/belt_block.c:210-269
 *   beltBlockEncr*   src/crypto/belt/belt_block.c:302-334
 *   CTR              src/crypto/belt/belt_ctr.c:27-135
 *   MAC              src/crypto/belt/belt_mac.c:47-203
 *   compress         src/crypto/belt/belt_compr.c:27-87
 *   hash             src/crypto/belt/belt_hash.c:43-190, belt_lcl.c:25-51
 */
#include "oracle.h"
#include "orc_threads.h"
#include <string.h>

/* ---------------------------------------------------------------- S-box --- */
static uint8_t g_H[256];
static pthread_once_t g_H_once = PTHREAD_ONCE_INIT;

static void gen_H(void)
{
    /* H[10] = 0, H[11] = 0x8E, then an 8-bit LFSR stepped 116 times per entry */
    g_H[10] = 0x00; g_H[11] = 0x8E;
    for (unsigned x = 12; x < 10 + 256; ++x) {
        unsigned t = g_H[(x - 1) % 256];
        for (int i = 0; i < 116; ++i) {
            unsigned par = __builtin_parity(t & 0x63);
            t = (t >> 1) | (par << 7);
        }
        g_H[x % 256] = (uint8_t)t;
    }
}

const uint8_t *orc_beltH(void)
{
    pthread_once(&g_H_once, gen_H);
    return g_H;
}

/* ------------------------------------------------------------ primitives --- */
static inline uint32_t rotl32(uint32_t x, unsigned r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t load32le(const uint8_t *p)
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
static inline void store32le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

static inline uint32_t G(const uint8_t *H, uint32_t x, unsigned r)
{
    uint32_t y = (uint32_t)H[x & 255] | (uint32_t)H[(x >> 8) & 255] << 8 |
                 (uint32_t)H[(x >> 16) & 255] << 16 | (uint32_t)H[x >> 24] << 24;
    return rotl32(y, r);
}

void orc_beltKeyExpand2(uint32_t key_[8], const uint8_t *key, size_t len)
{
    for (size_t i = 0; i < len / 4; ++i) key_[i] = load32le(key + 4 * i);
    if (len == 16) {
        for (int i = 0; i < 4; ++i) key_[4 + i] = key_[i];
    } else if (len == 24) {
        key_[6] = key_[0] ^ key_[1] ^ key_[2];
        key_[7] = key_[3] ^ key_[4] ^ key_[5];
    }
}

void orc_beltBlockEncr2(uint32_t block[4], const uint32_t K[8])
{
    const uint8_t *H = orc_beltH();
    uint32_t a = block[0], b = block[1], c = block[2], d = block[3], e, t;
    for (unsigned i = 1; i <= 8; ++i) {
        const unsigned base = 7 * i - 7;
#define KEY(j) K[(base + (j)) & 7]
        b ^= G(H, a + KEY(0), 5);
        c ^= G(H, d + KEY(1), 21);
        a -= G(H, b + KEY(2), 13);
        e = G(H, b + c + KEY(3), 21) ^ i;
        b += e;
        c -= e;
        d += G(H, c + KEY(4), 13);
        b ^= G(H, a + KEY(5), 21);
        c ^= G(H, d + KEY(6), 5);
#undef KEY
        /* (a,b,c,d) <- (b,d,a,c) */
        t = a; a = b; b = d; d = c; c = t;
    }
    /* output (b,d,a,c): belt_block.c:267-269 */
    block[0] = b; block[1] = d; block[2] = a; block[3] = c;
}

void orc_beltBlockEncr(uint8_t block[16], const uint32_t key[8])
{
    uint32_t w[4];
    for (int i = 0; i < 4; ++i) w[i] = load32le(block + 4 * i);
    orc_beltBlockEncr2(w, key);
    for (int i = 0; i < 4; ++i) store32le(block + 4 * i, w[i]);
}

/* ------------------------------------------------------------------- CTR --- */
static inline void ctr_add(uint32_t out[4], const uint32_t in[4], uint64_t add)
{
    /* 128-bit little-endian integer + 64-bit offset (belt_ctr.c:27-35 applied `add` times) */
    uint64_t lo = (uint64_t)in[0] | (uint64_t)in[1] << 32;
    uint64_t hi = (uint64_t)in[2] | (uint64_t)in[3] << 32;
    uint64_t nlo = lo + add;
    hi += (nlo < lo);
    out[0] = (uint32_t)nlo; out[1] = (uint32_t)(nlo >> 32);
    out[2] = (uint32_t)hi; out[3] = (uint32_t)(hi >> 32);
}

void orc_beltCTRStart(orc_belt_ctr_st *st, const uint8_t *key, size_t len, const uint8_t iv[16])
{
    orc_beltKeyExpand2(st->key, key, len);
    for (int i = 0; i < 4; ++i) st->ctr[i] = load32le(iv + 4 * i);
    orc_beltBlockEncr2(st->ctr, st->key);
    memset(st->block, 0, 16);
    st->reserved = 0;
}

static void ctr_next_gamma(orc_belt_ctr_st *st)
{
    uint32_t g[4];
    ctr_add(st->ctr, st->ctr, 1);
    memcpy(g, st->ctr, 16);
    orc_beltBlockEncr2(g, st->key);
    for (int i = 0; i < 4; ++i) store32le(st->block + 4 * i, g[i]);
}

void orc_beltCTRStepE(void *buf_, size_t count, orc_belt_ctr_st *st)
{
    uint8_t *buf = (uint8_t *)buf_;
    if (st->reserved) {                                   /* belt_ctr.c:70-83 */
        size_t take = st->reserved < count ? st->reserved : count;
        const uint8_t *g = st->block + 16 - st->reserved;
        for (size_t i = 0; i < take; ++i) buf[i] ^= g[i];
        st->reserved -= take; buf += take; count -= take;
        if (!count) return;
    }
    while (count >= 16) {                                 /* :85-97 */
        ctr_next_gamma(st);
        for (int i = 0; i < 16; ++i) buf[i] ^= st->block[i];
        buf += 16; count -= 16;
    }
    if (count) {                                          /* :98-110 */
        ctr_next_gamma(st);
        for (size_t i = 0; i < count; ++i) buf[i] ^= st->block[i];
        st->reserved = 16 - count;
    }
}

uint32_t orc_beltCTR(void *dest, const void *src, size_t count,
                     const uint8_t *key, size_t len, const uint8_t iv[16])
{
    orc_belt_ctr_st st;
    if (len != 16 && len != 24 && len != 32) return ORC_BAD_INPUT;
    orc_beltCTRStart(&st, key, len, iv);
    memmove(dest, src, count);
    orc_beltCTRStepE(dest, count, &st);
    return ORC_OK;
}

typedef struct { uint8_t *buf; const uint32_t *key; const uint32_t *ctr0; uint64_t first; } ctr_job;
static void ctr_range(void *ctx, size_t lo, size_t hi)
{
    ctr_job *j = (ctr_job *)ctx;
    for (size_t i = lo; i < hi; ++i) {
        uint32_t g[4];
        uint8_t *p = j->buf + 16 * i;
        ctr_add(g, j->ctr0, j->first + (uint64_t)i + 1);
        orc_beltBlockEncr2(g, j->key);
        for (int k = 0; k < 4; ++k) store32le(p + 4 * k, load32le(p + 4 * k) ^ g[k]);
    }
}
void orc_beltCTR_blocks(uint8_t *buf, size_t nblocks, const uint32_t key[8],
                        const uint32_t ctr0[4], uint64_t first, int nthreads)
{
    ctr_job j = {buf, key, ctr0, first};
    (void)orc_beltH();
    orc_parallel_for(nblocks, nthreads, ctr_range, &j);
}

/* ------------------------------------------------------------------- MAC --- */
void orc_beltMACStart(orc_belt_mac_st *st, const uint8_t *key, size_t len)
{
    orc_beltKeyExpand2(st->key, key, len);
    memset(st->s, 0, 16);
    memset(st->r, 0, 16);
    orc_beltBlockEncr2(st->r, st->key);
    memset(st->block, 0, 16);
    st->filled = 0;
}

static void mac_absorb_block(orc_belt_mac_st *st)
{
    for (int i = 0; i < 4; ++i) st->s[i] ^= load32le(st->block + 4 * i);
    orc_beltBlockEncr2(st->s, st->key);
}

void orc_beltMACStepA(const void *buf_, size_t count, orc_belt_mac_st *st)
{
    const uint8_t *buf = (const uint8_t *)buf_;
    /* one block of look-ahead: a full buffered block is absorbed only when more
       data arrives (belt_mac.c:58-99) */
    while (count) {
        if (st->filled == 16) { mac_absorb_block(st); st->filled = 0; }
        size_t take = 16 - st->filled;
        if (take > count) take = count;
        memcpy(st->block + st->filled, buf, take);
        st->filled += take; buf += take; count -= take;
    }
}

void orc_beltMACStepG(uint8_t mac[8], const orc_belt_mac_st *st)
{
    uint32_t m[4], x[4];
    uint8_t blk[16];
    const uint32_t *r = st->r;
    memcpy(blk, st->block, 16);
    if (st->filled == 16) {                               /* belt_mac.c:105-119 */
        for (int i = 0; i < 4; ++i) x[i] = load32le(blk + 4 * i);
        m[0] = st->s[0] ^ x[0] ^ r[1];
        m[1] = st->s[1] ^ x[1] ^ r[2];
        m[2] = st->s[2] ^ x[2] ^ r[3];
        m[3] = st->s[3] ^ x[3] ^ r[0] ^ r[1];
    } else {                                              /* :121-136 */
        blk[st->filled] = 0x80;
        memset(blk + st->filled + 1, 0, 16 - st->filled - 1);
        for (int i = 0; i < 4; ++i) x[i] = load32le(blk + 4 * i);
        m[0] = st->s[0] ^ x[0] ^ r[0] ^ r[3];
        m[1] = st->s[1] ^ x[1] ^ r[0];
        m[2] = st->s[2] ^ x[2] ^ r[1];
        m[3] = st->s[3] ^ x[3] ^ r[2];
    }
    orc_beltBlockEncr2(m, st->key);
    store32le(mac, m[0]);
    store32le(mac + 4, m[1]);
}

uint32_t orc_beltMAC(uint8_t mac[8], const void *src, size_t count,
                     const uint8_t *key, size_t len)
{
    orc_belt_mac_st st;
    if (len != 16 && len != 24 && len != 32) return ORC_BAD_INPUT;
    orc_beltMACStart(&st, key, len);
    orc_beltMACStepA(src, count, &st);
    orc_beltMACStepG(mac, &st);
    return ORC_OK;
}

/* ------------------------------------------------------- compress / hash --- */
/* sigma1(h, X) = E_X(h0 ^ h1) ^ h0 ^ h1 ; returns it in s1, updates h <- sigma2 */
static void compress_core(uint32_t s1[4], uint32_t h[8], const uint32_t X[8])
{
    uint32_t u[4], k1[8], k2[8], y0[4], y1[4];
    for (int i = 0; i < 4; ++i) u[i] = h[i] ^ h[4 + i];
    memcpy(s1, u, 16);
    orc_beltBlockEncr2(s1, X);
    for (int i = 0; i < 4; ++i) s1[i] ^= u[i];
    /* K1 = s1 || h1 ; K2 = ~s1 || h0 */
    for (int i = 0; i < 4; ++i) { k1[i] = s1[i]; k1[4 + i] = h[4 + i]; k2[i] = ~s1[i]; k2[4 + i] = h[i]; }
    memcpy(y0, X, 16);     orc_beltBlockEncr2(y0, k1);
    memcpy(y1, X + 4, 16); orc_beltBlockEncr2(y1, k2);
    for (int i = 0; i < 4; ++i) { h[i] = y0[i] ^ X[i]; h[4 + i] = y1[i] ^ X[4 + i]; }
}

void orc_beltCompr(uint32_t h[8], const uint32_t X[8])
{
    uint32_t s1[4];
    compress_core(s1, h, X);
}

uint32_t orc_beltHash(uint8_t hash[32], const void *src_, size_t count)
{
    const uint8_t *src = (const uint8_t *)src_;
    const uint8_t *H = orc_beltH();
    uint32_t ls[8], h[8], X[8], s1[4];
    uint8_t blk[32];
    memset(ls, 0, sizeof ls);
    /* len as a 128-bit LE bit count (belt_lcl.c:25-51) */
    {
        uint64_t bits_lo = (uint64_t)count << 3, bits_hi = (uint64_t)count >> 61;
        ls[0] = (uint32_t)bits_lo; ls[1] = (uint32_t)(bits_lo >> 32);
        ls[2] = (uint32_t)bits_hi; ls[3] = (uint32_t)(bits_hi >> 32);
    }
    for (int i = 0; i < 8; ++i) h[i] = load32le(H + 4 * i);
    while (count >= 32) {
        for (int i = 0; i < 8; ++i) X[i] = load32le(src + 4 * i);
        compress_core(s1, h, X);
        for (int i = 0; i < 4; ++i) ls[4 + i] ^= s1[i];
        src += 32; count -= 32;
    }
    if (count) {
        memset(blk, 0, 32);
        memcpy(blk, src, count);
        for (int i = 0; i < 8; ++i) X[i] = load32le(blk + 4 * i);
        compress_core(s1, h, X);
        for (int i = 0; i < 4; ++i) ls[4 + i] ^= s1[i];
    }
    orc_beltCompr(h, ls);
    for (int i = 0; i < 8; ++i) store32le(hash + 4 * i, h[i]);
    return ORC_OK;
}

/* ----------------------------------------- H4: bash512 + beltMAC / message --- */
typedef struct {
    const uint8_t *msgs; size_t msg_len; const uint8_t *key; size_t key_len;
    uint8_t *digests; uint8_t *tags;
} mixed_job;
static void mixed_range(void *ctx, size_t lo, size_t hi)
{
    mixed_job *j = (mixed_job *)ctx;
    for (size_t i = lo; i < hi; ++i) {
        const uint8_t *m = j->msgs + i * j->msg_len;
        orc_bashHash(j->digests + 64 * i, 256, m, j->msg_len);
        orc_beltMAC(j->tags + 8 * i, m, j->msg_len, j->key, j->key_len);
    }
}
void orc_bash512_beltMAC_batch(const uint8_t *msgs, size_t msg_len, size_t n,
                               const uint8_t *key, size_t key_len,
                               uint8_t *digests, uint8_t *tags, int nthreads)
{
    mixed_job j = {msgs, msg_len, key, key_len, digests, tags};
    (void)orc_beltH();
    orc_parallel_for(n, nthreads, mixed_range, &j);
}

/* ---- reference drivers (see bash_oracle.c) ---- */
typedef void (*ref_ctr_step_fn)(void *buf, size_t count, void *state);
typedef struct { uint8_t *buf; const uint32_t *key; const uint32_t *ctr0; uint64_t first; ref_ctr_step_fn f; } ref_ctr_job;
static void ref_ctr_range(void *ctx, size_t lo, size_t hi)
{
    ref_ctr_job *j = (ref_ctr_job *)ctx;
    /* a belt_ctr_st (belt_lcl.h:135-141) positioned at block `first + lo`: the reference
       has no seek, but its state is a flat POD the caller may construct */
    orc_belt_ctr_st st;
    memcpy(st.key, j->key, 32);
    ctr_add(st.ctr, j->ctr0, j->first + lo);
    memset(st.block, 0, 16);
    st.reserved = 0;
    j->f(j->buf + 16 * lo, 16 * (hi - lo), &st);
}
void orc_drive_ref_ctr(void *step_fn, uint8_t *buf, size_t nblocks, const uint32_t key[8],
                       const uint32_t ctr0[4], uint64_t first, int nthreads)
{
    ref_ctr_job j = {buf, key, ctr0, first, (ref_ctr_step_fn)step_fn};
    orc_parallel_for(nblocks, nthreads, ref_ctr_range, &j);
}

typedef uint32_t (*ref_hash_fn)(uint8_t *hash, size_t l, const void *src, size_t count);
typedef uint32_t (*ref_mac_fn)(uint8_t mac[8], const void *src, size_t count, const uint8_t *key, size_t len);
typedef struct {
    const uint8_t *msgs; size_t msg_len; const uint8_t *key; size_t key_len;
    uint8_t *digests; uint8_t *tags; ref_hash_fn h; ref_mac_fn m;
} ref_mixed_job;
static void ref_mixed_range(void *ctx, size_t lo, size_t hi)
{
    ref_mixed_job *j = (ref_mixed_job *)ctx;
    for (size_t i = lo; i < hi; ++i) {
        const uint8_t *p = j->msgs + i * j->msg_len;
        j->h(j->digests + 64 * i, 256, p, j->msg_len);
        j->m(j->tags + 8 * i, p, j->msg_len, j->key, j->key_len);
    }
}
void orc_drive_ref_mixed(void *hash_fn, void *mac_fn, const uint8_t *msgs, size_t msg_len, size_t n,
                         const uint8_t *key, size_t key_len, uint8_t *digests, uint8_t *tags, int nthreads)
{
    ref_mixed_job j = {msgs, msg_len, key, key_len, digests, tags, (ref_hash_fn)hash_fn, (ref_mac_fn)mac_fn};
    orc_parallel_for(n, nthreads, ref_mixed_range, &j);
}

/* ------------------------------------------- ECB / CBC --- */
/* D_K: the same round function with the subkeys in reverse order
   (belt_block.c:242-243 subkey_d, :286-295 macro D, :341-373 beltBlockDecr*) */
void orc_beltBlockDecr2(uint32_t block[4], const uint32_t K[8])
{
    const uint8_t *H = orc_beltH();
    uint32_t a = block[0], b = block[1], c = block[2], d = block[3], e, t;
    for (unsigned i = 8; i >= 1; --i) {
        const unsigned base = 7 * i - 1;
#define KEY(j) K[(base - (j)) & 7]
        b ^= G(H, a + KEY(0), 5);
        c ^= G(H, d + KEY(1), 21);
        a -= G(H, b + KEY(2), 13);
        e = G(H, b + c + KEY(3), 21) ^ i;
        b += e;
        c -= e;
        d += G(H, c + KEY(4), 13);
        b ^= G(H, a + KEY(5), 21);
        c ^= G(H, d + KEY(6), 5);
#undef KEY
        /* (a,b,c,d) <- (c,a,d,b) */
        t = a; a = c; c = d; d = b; b = t;
    }
    /* output (c,a,d,b): belt_block.c:293-295 */
    block[0] = c; block[1] = a; block[2] = d; block[3] = b;
}

static void blk_load(uint32_t w[4], const uint8_t *p) { for (int i = 0; i < 4; ++i) w[i] = load32le(p + 4 * i); }
static void blk_store(uint8_t *p, const uint32_t w[4]) { for (int i = 0; i < 4; ++i) store32le(p + 4 * i, w[i]); }
static void blk_crypt(uint8_t *p, const uint32_t key[8], int decr)
{
    uint32_t w[4];
    blk_load(w, p);
    if (decr) orc_beltBlockDecr2(w, key); else orc_beltBlockEncr2(w, key);
    blk_store(p, w);
}

/* beltECBEncr / beltECBDecr with ciphertext stealing (belt_ecb.c:52-159) */
uint32_t orc_beltECB(void *dest, const void *src, size_t count, const uint8_t *key, size_t len, int decr)
{
    uint32_t K[8];
    uint8_t *buf = (uint8_t *)dest, blk[16];
    if (count < 16 || (len != 16 && len != 24 && len != 32)) return ORC_BAD_INPUT;
    orc_beltKeyExpand2(K, key, len);
    memmove(dest, src, count);
    while (count >= 16) { blk_crypt(buf, K, decr); buf += 16; count -= 16; }
    if (count) {
        memcpy(blk, buf, count);
        memcpy(blk + count, buf - 16 + count, 16 - count);
        blk_crypt(blk, K, decr);
        memcpy(buf, buf - 16, count);
        memcpy(buf - 16, blk, 16);
    }
    return ORC_OK;
}

/* s <- s * x in GF(2^128) = GF(2)[x] / (x^128 + x^7 + x^2 + x + 1); s is the 128-bit little-endian
   value s[0] + 2^32 s[1] + ... (beltBlockMulC, belt_lcl.c:99-108) */
static void gf128_double(uint32_t s[4])
{
    const uint32_t out = s[3] >> 31;                 /* coefficient of x^127 leaves the block */
    int i;
    for (i = 3; i > 0; --i) s[i] = (s[i] << 1) | (s[i - 1] >> 31);
    s[0] = (s[0] << 1) ^ (out ? 0x87u : 0u);        /* x^128 = x^7 + x^2 + x + 1 */
}

/* the block loop of beltBDEStepE / StepD (belt_bde.c:51-85) from a given tweak state s (updated in place):
   what one rank does with its shard once s has been advanced to the shard's first block */
void orc_beltBDE_blocks(void *buf_, size_t nblocks, const uint32_t K[8], uint32_t s[4], int decr)
{
    uint8_t *buf = (uint8_t *)buf_;
    uint32_t w[4];
    int i;
    for (; nblocks; --nblocks, buf += 16) {
        gf128_double(s);
        blk_load(w, buf);
        for (i = 0; i < 4; ++i) w[i] ^= s[i];
        if (decr) orc_beltBlockDecr2(w, K); else orc_beltBlockEncr2(w, K);
        for (i = 0; i < 4; ++i) w[i] ^= s[i];
        blk_store(buf, w);
    }
}

/* beltBDEEncr / beltBDEDecr (belt_bde.c:40-133): s <- E_K(iv); for every block
   s <- s * x, Y = E_K(X ^ s) ^ s (decryption: D_K).  Whole blocks only. */
uint32_t orc_beltBDE(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     const uint8_t iv[16], int decr)
{
    uint32_t K[8], s[4];
    if (count < 16 || count % 16 || (len != 16 && len != 24 && len != 32)) return ORC_BAD_INPUT;
    orc_beltKeyExpand2(K, key, len);
    blk_load(s, iv);
    orc_beltBlockEncr2(s, K);
    memmove(dest, src, count);
    orc_beltBDE_blocks(dest, count / 16, K, s, decr);
    return ORC_OK;
}

/* ---- belt-wbl on whole blocks and belt-sde (belt_wbl.c:58-152, belt_sde.c:38-121) ----
   Wide-block cipher on n >= 2 blocks r_1..r_n.  Encryption round i = 1..2n:
       s = r_1 ^ ... ^ r_{n-1};   (r_1, .., r_n) <- (r_2, .., r_{n-1}, r_n ^ E_K(s) ^ <i>, s)
   with <i> the round number as a 64-bit little-endian integer in the first 8 bytes of the block.
   Decryption runs the rounds i = 2n..1 backwards.  Written from the definition (the sum is
   recomputed and the blocks really move every round), not in the reference's rolling-sum form. */
static void wbl_round_word(uint32_t e[4], uint64_t i) { e[0] ^= (uint32_t)i; e[1] ^= (uint32_t)(i >> 32); }
uint32_t orc_beltWBL(void *buf_, size_t nblocks, const uint32_t K[8], int decr)
{
    uint8_t *buf = (uint8_t *)buf_;
    const size_t n = nblocks;
    uint64_t i;
    size_t j;
    int k;
    if (n < 2) return ORC_BAD_INPUT;
    if (!decr) {
        for (i = 1; i <= 2 * (uint64_t)n; ++i) {
            uint32_t s[4] = {0, 0, 0, 0}, e[4], w[4];
            uint8_t sb[16];
            for (j = 0; j + 1 < n; ++j) { blk_load(w, buf + 16 * j); for (k = 0; k < 4; ++k) s[k] ^= w[k]; }
            blk_store(sb, s);
            for (k = 0; k < 4; ++k) e[k] = s[k];
            orc_beltBlockEncr2(e, K);
            wbl_round_word(e, i);
            blk_load(w, buf + 16 * (n - 1));
            for (k = 0; k < 4; ++k) w[k] ^= e[k];
            memmove(buf, buf + 16, 16 * (n - 2));            /* r_2 .. r_{n-1} move down */
            blk_store(buf + 16 * (n - 2), w);                  /* r_n ^ E(s) ^ <i> */
            memcpy(buf + 16 * (n - 1), sb, 16);                /* s */
        }
    } else {
        for (i = 2 * (uint64_t)n; i >= 1; --i) {
            uint32_t s[4], e[4], w[4], r1[4];
            blk_load(s, buf + 16 * (n - 1));                   /* s = r_n */
            for (k = 0; k < 4; ++k) e[k] = s[k];
            orc_beltBlockEncr2(e, K);
            wbl_round_word(e, i);
            blk_load(w, buf + 16 * (n - 2));                   /* old r_n = r_{n-1} ^ E(s) ^ <i> */
            for (k = 0; k < 4; ++k) w[k] ^= e[k];
            memmove(buf + 16, buf, 16 * (n - 2));              /* r_1 .. r_{n-2} become r_2 .. r_{n-1} */
            blk_store(buf + 16 * (n - 1), w);
            for (k = 0; k < 4; ++k) r1[k] = s[k];              /* old r_1 = s ^ r_2 ^ .. ^ r_{n-1} */
            for (j = 1; j + 1 < n; ++j) { blk_load(w, buf + 16 * j); for (k = 0; k < 4; ++k) r1[k] ^= w[k]; }
            blk_store(buf, r1);
        }
    }
    return ORC_OK;
}
/* beltSDEEncr / beltSDEDecr: XEX around belt-wbl with the tweak E_K(iv) on the first block (belt_sde.c:47-71) */
uint32_t orc_beltSDE(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     const uint8_t iv[16], int decr)
{
    uint32_t K[8], s[4], w[4];
    uint8_t *buf = (uint8_t *)dest;
    int k;
    if (count % 16 || count < 32 || (len != 16 && len != 24 && len != 32)) return ORC_BAD_INPUT;
    orc_beltKeyExpand2(K, key, len);
    blk_load(s, iv);
    orc_beltBlockEncr2(s, K);
    memmove(dest, src, count);
    blk_load(w, buf); for (k = 0; k < 4; ++k) w[k] ^= s[k]; blk_store(buf, w);
    orc_beltWBL(buf, count / 16, K, decr);
    blk_load(w, buf); for (k = 0; k < 4; ++k) w[k] ^= s[k]; blk_store(buf, w);
    return ORC_OK;
}

/* ---- belt-dwp: CTR encryption + polynomial MAC over GF(2^128) ------------
   State machine of belt_dwp.c:27-196.  The authenticator is t <- (t ^ X) * r for every 16-byte
   block X of the open data, then of the critical data (each zero-padded to whole blocks), then of
   the block <bit length of open data>_64 || <bit length of critical data>_64; mac = E_K(t)[0..8). */
static void gf128_mul(uint32_t c[4], const uint32_t a_[4], const uint32_t b[4])     /* belt_lcl.c:119-132 */
{
    uint32_t a[4] = {a_[0], a_[1], a_[2], a_[3]}, r[4] = {0, 0, 0, 0};
    int i, j;
    for (i = 0; i < 128; ++i) {
        if ((b[i >> 5] >> (i & 31)) & 1u) for (j = 0; j < 4; ++j) r[j] ^= a[j];
        gf128_double(a);
    }
    for (j = 0; j < 4; ++j) c[j] = r[j];
}
static void dwp_absorb(uint32_t t[4], const uint8_t block[16], const uint32_t r[4])
{
    uint32_t x[4];
    int i;
    blk_load(x, block);
    for (i = 0; i < 4; ++i) x[i] ^= t[i];
    gf128_mul(t, x, r);
}
void orc_beltDWPStart(orc_belt_dwp_st *st, const uint8_t *key, size_t len, const uint8_t iv[16])
{
    int i;
    orc_beltCTRStart(&st->ctr, key, len, iv);                   /* ctr = E_K(iv) */
    for (i = 0; i < 4; ++i) st->r[i] = st->ctr.ctr[i];
    orc_beltBlockEncr2(st->r, st->ctr.key);                     /* r = E_K(ctr), belt_dwp.c:52-54 */
    blk_load(st->t, orc_beltH());                               /* t = H[0..16), :59 */
    st->bits_open = st->bits_crit = 0;
    st->filled = 0;
}
void orc_beltDWPStepE(void *buf, size_t count, orc_belt_dwp_st *st) { orc_beltCTRStepE(buf, count, &st->ctr); }
static void dwp_feed(orc_belt_dwp_st *st, const uint8_t *p, size_t count)
{
    while (count) {
        size_t take = 16 - st->filled;
        if (take > count) take = count;
        memcpy(st->block + st->filled, p, take);
        st->filled += take; p += take; count -= take;
        if (st->filled == 16) { dwp_absorb(st->t, st->block, st->r); st->filled = 0; }
    }
}
static void dwp_flush(orc_belt_dwp_st *st)
{
    if (st->filled) {
        memset(st->block + st->filled, 0, 16 - st->filled);
        dwp_absorb(st->t, st->block, st->r);
        st->filled = 0;
    }
}
void orc_beltDWPStepI(const void *buf, size_t count, orc_belt_dwp_st *st)       /* belt_dwp.c:68-107 */
{
    st->bits_open += (uint64_t)count * 8;
    dwp_feed(st, (const uint8_t *)buf, count);
}
void orc_beltDWPStepA(const void *buf, size_t count, orc_belt_dwp_st *st)       /* belt_dwp.c:109-155 */
{
    if (count && st->bits_crit == 0) dwp_flush(st);        /* open data ends on a block boundary */
    st->bits_crit += (uint64_t)count * 8;
    dwp_feed(st, (const uint8_t *)buf, count);
}
void orc_beltDWPStepG(uint8_t mac[8], const orc_belt_dwp_st *st)                /* belt_dwp.c:162-189 */
{
    orc_belt_dwp_st c = *st;                               /* the state itself is not disturbed */
    uint8_t lenblk[16], out[16];
    int i;
    dwp_flush(&c);
    for (i = 0; i < 8; ++i) {
        lenblk[i] = (uint8_t)(c.bits_open >> (8 * i));
        lenblk[8 + i] = (uint8_t)(c.bits_crit >> (8 * i));
    }
    dwp_absorb(c.t, lenblk, c.r);
    orc_beltBlockEncr2(c.t, c.ctr.key);
    blk_store(out, c.t);
    memcpy(mac, out, 8);
}
uint32_t orc_beltDWPWrap(void *dest, uint8_t mac[8], const void *src1, size_t count1, const void *src2,
                         size_t count2, const uint8_t *key, size_t len, const uint8_t iv[16])
{
    orc_belt_dwp_st st;
    if (len != 16 && len != 24 && len != 32) return ORC_BAD_INPUT;
    orc_beltDWPStart(&st, key, len, iv);
    orc_beltDWPStepI(src2, count2, &st);
    memmove(dest, src1, count1);
    orc_beltDWPStepE(dest, count1, &st);
    orc_beltDWPStepA(dest, count1, &st);
    orc_beltDWPStepG(mac, &st);
    return ORC_OK;
}
uint32_t orc_beltDWPUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                           const uint8_t mac[8], const uint8_t *key, size_t len, const uint8_t iv[16])
{
    orc_belt_dwp_st st;
    uint8_t m[8];
    if (len != 16 && len != 24 && len != 32) return ORC_BAD_INPUT;
    orc_beltDWPStart(&st, key, len, iv);
    orc_beltDWPStepI(src2, count2, &st);
    orc_beltDWPStepA(src1, count1, &st);
    orc_beltDWPStepG(m, &st);
    if (memcmp(m, mac, 8)) return ORC_BAD_MAC;             /* ERR_BAD_MAC, nothing is decrypted */
    memmove(dest, src1, count1);
    orc_beltDWPStepE(dest, count1, &st);
    return ORC_OK;
}

/* ---- belt-che (belt_che.c:27-319): the same authenticator as belt-dwp with
   r = E_K(iv), and the keystream gamma_i = E_K(s_i), s_0 = r, s_i = s_{i-1} * x ^ 1 in GF(2^128);
   unused gamma bytes of a partial block are kept for the next call (belt_che.c:69-112). */
void orc_beltCHEStart(orc_belt_che_st *st, const uint8_t *key, size_t len, const uint8_t iv[16])
{
    int i;
    memset(st, 0, sizeof *st);
    orc_beltKeyExpand2(st->mac.ctr.key, key, len);
    blk_load(st->mac.r, iv);
    orc_beltBlockEncr2(st->mac.r, st->mac.ctr.key);             /* r = E_K(iv), :54-56 */
    for (i = 0; i < 4; ++i) st->s[i] = st->mac.r[i];            /* s = r */
    blk_load(st->mac.t, orc_beltH());
}
void orc_beltCHEStepE(void *buf_, size_t count, orc_belt_che_st *st)
{
    uint8_t *buf = (uint8_t *)buf_;
    while (count) {
        if (st->reserved == 0) {
            uint32_t g[4];
            int i;
            gf128_double(st->s);
            st->s[0] ^= 1u;
            for (i = 0; i < 4; ++i) g[i] = st->s[i];
            orc_beltBlockEncr2(g, st->mac.ctr.key);
            blk_store(st->gamma, g);
            st->reserved = 16;
        }
        *buf++ ^= st->gamma[16 - st->reserved];
        --st->reserved;
        --count;
    }
}
/* whole blocks of the belt-che keystream from a given state s (updated in place), belt_che.c:86-98 */
void orc_beltCHE_blocks(void *buf_, size_t nblocks, const uint32_t K[8], uint32_t s[4])
{
    uint8_t *buf = (uint8_t *)buf_;
    uint32_t g[4], w[4];
    int i;
    for (; nblocks; --nblocks, buf += 16) {
        gf128_double(s);
        s[0] ^= 1u;
        for (i = 0; i < 4; ++i) g[i] = s[i];
        orc_beltBlockEncr2(g, K);
        blk_load(w, buf);
        for (i = 0; i < 4; ++i) w[i] ^= g[i];
        blk_store(buf, w);
    }
}
void orc_beltCHEStepI(const void *buf, size_t count, orc_belt_che_st *st) { orc_beltDWPStepI(buf, count, &st->mac); }
void orc_beltCHEStepA(const void *buf, size_t count, orc_belt_che_st *st) { orc_beltDWPStepA(buf, count, &st->mac); }
void orc_beltCHEStepG(uint8_t mac[8], const orc_belt_che_st *st) { orc_beltDWPStepG(mac, &st->mac); }
uint32_t orc_beltCHEWrap(void *dest, uint8_t mac[8], const void *src1, size_t count1, const void *src2,
                         size_t count2, const uint8_t *key, size_t len, const uint8_t iv[16])
{
    orc_belt_che_st st;
    if (len != 16 && len != 24 && len != 32) return ORC_BAD_INPUT;
    orc_beltCHEStart(&st, key, len, iv);
    orc_beltCHEStepI(src2, count2, &st);
    memmove(dest, src1, count1);
    orc_beltCHEStepE(dest, count1, &st);
    orc_beltCHEStepA(dest, count1, &st);
    orc_beltCHEStepG(mac, &st);
    return ORC_OK;
}
uint32_t orc_beltCHEUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                           const uint8_t mac[8], const uint8_t *key, size_t len, const uint8_t iv[16])
{
    orc_belt_che_st st;
    uint8_t m[8];
    if (len != 16 && len != 24 && len != 32) return ORC_BAD_INPUT;
    orc_beltCHEStart(&st, key, len, iv);
    orc_beltCHEStepI(src2, count2, &st);
    orc_beltCHEStepA(src1, count1, &st);
    orc_beltCHEStepG(m, &st);
    if (memcmp(m, mac, 8)) return ORC_BAD_MAC;
    memmove(dest, src1, count1);
    orc_beltCHEStepE(dest, count1, &st);
    return ORC_OK;
}

/* beltCBCEncr / beltCBCDecr with ciphertext stealing (belt_cbc.c:63-193) */
uint32_t orc_beltCBC(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     const uint8_t iv[16], int decr)
{
    uint32_t K[8];
    uint8_t *buf = (uint8_t *)dest, chain[16], t[16];
    if (count < 16 || (len != 16 && len != 24 && len != 32)) return ORC_BAD_INPUT;
    orc_beltKeyExpand2(K, key, len);
    memmove(dest, src, count);
    memcpy(chain, iv, 16);
    if (!decr) {
        while (count >= 16) {
            for (int i = 0; i < 16; ++i) chain[i] ^= buf[i];
            blk_crypt(chain, K, 0);
            memcpy(buf, chain, 16);
            buf += 16; count -= 16;
        }
        if (count) {
            for (size_t i = 0; i < count; ++i) t[i] = buf[i] ^ chain[i];
            memcpy(t + count, buf - 16 + count, 16 - count);
            blk_crypt(t, K, 0);
            memcpy(buf, buf - 16, count);
            memcpy(buf - 16, t, 16);
        }
    } else {
        while (count >= 32 || count == 16) {
            memcpy(t, buf, 16);
            blk_crypt(t, K, 1);
            for (int i = 0; i < 16; ++i) t[i] ^= chain[i];
            memcpy(chain, buf, 16);
            memcpy(buf, t, 16);
            buf += 16; count -= 16;
        }
        if (count) {
            const size_t r = count - 16;
            memcpy(t, buf, 16);
            blk_crypt(t, K, 1);
            for (size_t i = 0; i < r; ++i) { uint8_t x = t[i]; t[i] = buf[16 + i]; buf[16 + i] = x; }
            for (size_t i = 0; i < r; ++i) buf[16 + i] ^= t[i];
            blk_crypt(t, K, 1);
            for (int i = 0; i < 16; ++i) buf[i] = t[i] ^ chain[i];
        }
    }
    return ORC_OK;
}

/* reference driver for the 8f-1 modes: fn = beltECBEncr / beltECBDecr (iv == NULL) or
   beltCBCDecr (iv given; slice k chains from the ciphertext block before it) */
typedef uint32_t (*ref_ecb_fn)(void *dest, const void *src, size_t count, const uint8_t *key, size_t len);
typedef uint32_t (*ref_cbc_fn)(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                               const uint8_t *iv);
typedef struct { const uint8_t *src; uint8_t *dst; const uint8_t *key; size_t klen; const uint8_t *iv; void *fn; } ref_mode_job;
static void ref_mode_range(void *ctx, size_t lo, size_t hi)
{
    ref_mode_job *j = (ref_mode_job *)ctx;
    if (hi == lo) return;
    if (!j->iv) ((ref_ecb_fn)j->fn)(j->dst + 16 * lo, j->src + 16 * lo, 16 * (hi - lo), j->key, j->klen);
    else ((ref_cbc_fn)j->fn)(j->dst + 16 * lo, j->src + 16 * lo, 16 * (hi - lo), j->key, j->klen,
                             lo ? j->src + 16 * (lo - 1) : j->iv);
}
void orc_drive_ref_mode(void *fn, const uint8_t *src, uint8_t *dst, size_t nblocks, const uint8_t *key,
                        size_t klen, const uint8_t *iv, int nthreads)
{
    ref_mode_job j = {src, dst, key, klen, iv, fn};
    orc_parallel_for(nblocks, nthreads, ref_mode_range, &j);
}
