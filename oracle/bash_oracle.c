/*
 * bash_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of STB 34.101.77 bash-f and the bash hash sponge, written in
 * the "explicit" form (S-layer on the 3x8 matrix, then the word permutation,
 * then the round constant) rather than bee2's index-renaming macros.
 *
 * Follows:
 *   bashS      src/crypto/bash/bash_f64.c:32-44      (column S-box)
 *   bashR/P    src/crypto/bash/bash_f64.c:100-134    (round + word permutation)
 *   constants  src/crypto/bash/bash_f64.c:50-59      (LFSR recurrence, computed here)
 *   bashF      src/crypto/bash/bash_f64.c:174-187
 *   bashHash*  src/crypto/bash/bash_hash.c:38-137
 */
#include "oracle.h"
#include "orc_threads.h"
#include <string.h>

static inline uint64_t rotl64(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }

static inline uint64_t load64le(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
static inline void store64le(uint8_t *p, uint64_t v)
{
    for (int i = 0; i < 8; ++i) { p[i] = (uint8_t)v; v >>= 8; }
}

/* rotation amounts of column j: (8,53,14,1) * 7^j mod 64  (bash_f64.c:126-133) */
static void rot_params(unsigned j, unsigned *m1, unsigned *n1, unsigned *m2, unsigned *n2)
{
    unsigned a = 8, b = 53, c = 14, d = 1;
    for (unsigned k = 0; k < j; ++k) {
        a = (a * 7) % 64; b = (b * 7) % 64; c = (c * 7) % 64; d = (d * 7) % 64;
    }
    *m1 = a; *n1 = b; *m2 = c; *n2 = d;
}

/* new_row0 = pi1(old_row1), new_row1 = pi2(old_row2), new_row2 = pi0(old_row0) */
static const uint8_t PI0[8] = {6, 3, 0, 5, 2, 7, 4, 1};
static const uint8_t PI1[8] = {7, 2, 1, 4, 3, 6, 5, 0};
static const uint8_t PI2[8] = {1, 0, 3, 2, 5, 4, 7, 6};

static void bashF_words(uint64_t S[24])
{
    uint64_t C = 0x3BF5080AC8BA94B1ull;            /* C_1, bash_f64.c:50-59 */
    unsigned m1[8], n1[8], m2[8], n2[8];
    for (unsigned j = 0; j < 8; ++j) rot_params(j, &m1[j], &n1[j], &m2[j], &n2[j]);

    for (int round = 0; round < 24; ++round) {
        uint64_t N[24];
        for (unsigned j = 0; j < 8; ++j) {
            uint64_t w0 = S[j], w1 = S[8 + j], w2 = S[16 + j];
            uint64_t u0 = w0 ^ w1 ^ w2;
            uint64_t t = w1 ^ rotl64(u0, n1[j]);
            uint64_t u1 = t ^ rotl64(w0, m1[j]);
            uint64_t u2 = w2 ^ rotl64(w2, m2[j]) ^ rotl64(t, n2[j]);
            S[j] = u0 ^ (~u2 | u1);
            S[8 + j] = u1 ^ (u0 | u2);
            S[16 + j] = u2 ^ (u0 & u1);
        }
        for (unsigned k = 0; k < 8; ++k) {
            N[k] = S[8 + PI1[k]];
            N[8 + k] = S[16 + PI2[k]];
            N[16 + k] = S[PI0[k]];
        }
        N[23] ^= C;
        C = (C >> 1) ^ (0xDC2BE1997FE0D8AEull & (0 - (C & 1)));
        memcpy(S, N, sizeof N);
    }
}

void orc_bashF(uint8_t block[192])
{
    uint64_t S[24];
    for (int i = 0; i < 24; ++i) S[i] = load64le(block + 8 * i);
    bashF_words(S);
    for (int i = 0; i < 24; ++i) store64le(block + 8 * i, S[i]);
}

typedef struct { uint8_t *states; } bashF_job;
static void bashF_range(void *ctx, size_t lo, size_t hi)
{
    bashF_job *j = (bashF_job *)ctx;
    for (size_t i = lo; i < hi; ++i) orc_bashF(j->states + 192 * i);
}
void orc_bashF_batch(uint8_t *states, size_t n, int nthreads)
{
    bashF_job j = {states};
    orc_parallel_for(n, nthreads, bashF_range, &j);
}

/* ---- sponge hash: bash_hash.c:38-102 ---- */
void orc_bashHashStart(orc_bash_hash_st *st, size_t l)
{
    memset(st->s, 0, 192);
    st->s[192 - 8] = (uint8_t)(l / 4);
    st->buf_len = 192 - l / 2;
    st->pos = 0;
}

void orc_bashHashStepH(const uint8_t *buf, size_t count, orc_bash_hash_st *st)
{
    while (count) {
        size_t room = st->buf_len - st->pos;
        size_t take = count < room ? count : room;
        memcpy(st->s + st->pos, buf, take);
        st->pos += take; buf += take; count -= take;
        if (st->pos == st->buf_len) { orc_bashF(st->s); st->pos = 0; }
    }
}

void orc_bashHashStepG(uint8_t *hash, size_t hash_len, const orc_bash_hash_st *st)
{
    uint8_t s1[192];
    memcpy(s1, st->s, 192);
    /* pos == 0 gives a whole extra padding block (bash_hash.c:89-100) */
    memset(s1 + st->pos, 0, st->buf_len - st->pos);
    s1[st->pos] = 0x40;
    orc_bashF(s1);
    memcpy(hash, s1, hash_len);
}

uint32_t orc_bashHash(uint8_t *hash, size_t l, const uint8_t *src, size_t count)
{
    orc_bash_hash_st st;
    if (l == 0 || l % 16 != 0 || l > 256) return ORC_BAD_PARAMS;
    orc_bashHashStart(&st, l);
    orc_bashHashStepH(src, count, &st);
    orc_bashHashStepG(hash, l / 4, &st);
    return ORC_OK;
}

/* ---- splitmix64 synthetic generator ---- */
void orc_fill_splitmix64(uint8_t *buf, size_t nbytes, uint64_t seed)
{
    size_t nw = nbytes / 8;
    for (size_t i = 0; i < nw; ++i) {
        uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        store64le(buf + 8 * i, z);
    }
    if (nbytes % 8) {                 /* ragged tail: the leading bytes of the next word (the generator the
                                         fixtures name: "splitmix64 LE words", tools/make_golden.py) */
        uint8_t last[8];
        uint64_t z = seed + (uint64_t)nw * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        store64le(last, z);
        memcpy(buf + 8 * nw, last, nbytes % 8);
    }
}

/* ---- drivers for timing the REFERENCE itself (oracle/_ref) on many host threads ----
 * bench.py loads libbee2ref*.so with ctypes and hands the function pointers in; the
 * loops live here so that no per-call Python overhead is timed. */
typedef void (*ref_bashF_fn)(uint8_t *block, void *stack);
typedef struct { uint8_t *states; ref_bashF_fn f; } ref_bashF_job;
static void ref_bashF_range(void *ctx, size_t lo, size_t hi)
{
    ref_bashF_job *j = (ref_bashF_job *)ctx;
    uint8_t stack[256];                       /* bashF_deep() <= 200 on every platform */
    for (size_t i = lo; i < hi; ++i) j->f(j->states + 192 * i, stack);
}
void orc_drive_ref_bashF(void *fn, uint8_t *states, size_t n, int nthreads)
{
    ref_bashF_job j = {states, (ref_bashF_fn)fn};
    orc_parallel_for(n, nthreads, ref_bashF_range, &j);
}
