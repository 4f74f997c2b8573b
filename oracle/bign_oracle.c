/*
 * bign_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of STB 34.101.45 signature verification on bign-curve256v1
 * (bee2 names the same curve bign128: security level l = 128).
 *
 * Follows:
 *   bign128Verify   src/crypto/bign/bign128.c:151-153,177-185 (fixed belt-hash OID DER)
 *   bignVerifyEc    src/crypto/bign/bign_sign.c:268-347       (checks, s1+H mod q, s0+2^l, hash tail)
 *   ecAddMulA       src/math/ec.c:1183-1273   (interleaved width-5 NAF over the two scalars)
 *   ecPreSO         src/math/ec.c:164-196     (odd multiples 1,3,..,15)
 *   Jacobian ops    src/math/ecp/ecp_j.c:241-299 (dbl, a = -3), :397-497 (add), :104-133 (to affine)
 *                   -- exceptional cases (O operands, P = +-Q, y = 0) handled as there
 *   GF(p)           src/math/zm.c:214-263 (Crandall mul/sqr), src/math/zz/zz_red.c:71-105,
 *                   src/math/gfp.c:33-44 (inverse = a^(p-2))
 *   constants       src/crypto/bign/bign_params.c:36-73 (STB 34.101.45 annex B data)
 *
 * The point formulas are the textbook EFD ones (dbl-2001-b, add-2007-bl); the
 * Jacobian representative may differ from bee2's by a scalar factor, the affine
 * result -- the only thing the algorithm outputs -- cannot.
 */
#include "oracle.h"
#include "orc_threads.h"
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;      /* canonical residue in [0, p) */
typedef struct { fe X, Y, Z; } jac;         /* O  <=>  Z == 0 */

#define CRANDALL_C 189u                     /* p = 2^256 - 189 */
static const fe FE_P = {{0xFFFFFFFFFFFFFF43ull, ~0ull, ~0ull, ~0ull}};

/* STB 34.101.45 annex B.1 (bign-curve256v1), little-endian octets */
static const uint8_t Q_ORDER[32] = {
    0x07, 0x66, 0x3D, 0x26, 0x99, 0xBF, 0x5A, 0x7E, 0xFC, 0x4D, 0xFB, 0x0D, 0xD6, 0x8E, 0x5C, 0xD9,
    0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
static const uint8_t G_Y[32] = {
    0x93, 0x6A, 0x51, 0x04, 0x18, 0xCF, 0x29, 0x1E, 0x52, 0xF6, 0x08, 0xC4, 0x66, 0x39, 0x91, 0x78,
    0x5D, 0x83, 0xD6, 0x51, 0xA3, 0xC9, 0xE4, 0x5C, 0x9F, 0xD6, 0x16, 0xFB, 0x3C, 0xFC, 0xF7, 0x6B};
/* DER(1.2.112.0.2.0.34.101.31.81) = belt-hash, bign128.c:151-153 */
static const uint8_t OID_BELT_HASH[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};

/* ------------------------------------------------------------- integers --- */
static void u256_from_le(uint64_t w[4], const uint8_t *p)
{
    for (int i = 0; i < 4; ++i) {
        uint64_t v = 0;
        for (int k = 7; k >= 0; --k) v = (v << 8) | p[8 * i + k];
        w[i] = v;
    }
}
static void u256_to_le(uint8_t *p, const uint64_t w[4])
{
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 8; ++k) p[8 * i + k] = (uint8_t)(w[i] >> (8 * k));
}
static int u256_cmp(const uint64_t a[4], const uint64_t b[4])
{
    for (int i = 3; i >= 0; --i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
static uint64_t u256_add(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
{
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t u256_sub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
{
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t t = a[i] - b[i];
        uint64_t b2 = (a[i] < b[i]) | ((t < borrow) ? 1u : 0u);
        r[i] = t - borrow;
        borrow = b2;
    }
    return borrow;
}

/* ----------------------------------------------------------------- GF(p) --- */
static int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return u256_cmp(a->v, b->v) == 0; }

static void fe_add(fe *r, const fe *a, const fe *b)
{
    uint64_t t[4], carry = u256_add(t, a->v, b->v);
    if (carry || u256_cmp(t, FE_P.v) >= 0) u256_sub(t, t, FE_P.v);
    memcpy(r->v, t, sizeof t);
}
static void fe_sub(fe *r, const fe *a, const fe *b)
{
    uint64_t t[4];
    if (u256_sub(t, a->v, b->v)) u256_add(t, t, FE_P.v);
    memcpy(r->v, t, sizeof t);
}
static void fe_dbl(fe *r, const fe *a) { fe_add(r, a, a); }

/* 4x4 schoolbook (zz_mul.c:82-105) then Crandall fold by c = 189 (zz_red.c:71-105) */
static void fe_mul(fe *r, const fe *a, const fe *b)
{
    uint64_t w[8] = {0};
    for (int i = 0; i < 4; ++i) {
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            carry += (u128)a->v[i] * b->v[j] + w[i + j];
            w[i + j] = (uint64_t)carry;
            carry >>= 64;
        }
        w[i + 4] = (uint64_t)carry;
    }
    /* lo + c * hi  ->  5 words */
    uint64_t t[4];
    u128 acc = 0;
    for (int i = 0; i < 4; ++i) {
        acc += (u128)w[4 + i] * CRANDALL_C + w[i];
        t[i] = (uint64_t)acc;
        acc >>= 64;
    }
    /* top word (< 190) folds once more; a final carry folds as +c */
    acc = (u128)(uint64_t)acc * CRANDALL_C;
    for (int i = 0; i < 4; ++i) { acc += t[i]; t[i] = (uint64_t)acc; acc >>= 64; }
    if (acc) {
        acc = CRANDALL_C;
        for (int i = 0; i < 4; ++i) { acc += t[i]; t[i] = (uint64_t)acc; acc >>= 64; }
    }
    if (u256_cmp(t, FE_P.v) >= 0) u256_sub(t, t, FE_P.v);
    memcpy(r->v, t, sizeof t);
}
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }

/* a^(p-2), p - 2 = (2^248 - 1) * 2^8 + 0x41   (gfp.c:33-44: Fermat inverse) */
static void fe_inv(fe *r, const fe *a)
{
    fe x = *a, acc;
    /* plain left-to-right square-and-multiply over the 256 exponent bits */
    uint64_t e[4] = {FE_P.v[0] - 2, FE_P.v[1], FE_P.v[2], FE_P.v[3]};
    acc.v[0] = 1; acc.v[1] = acc.v[2] = acc.v[3] = 0;
    for (int i = 255; i >= 0; --i) {
        fe_sqr(&acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &x);
    }
    *r = acc;
}

/* -------------------------------------------------------------- Jacobian --- */
static void jac_set_inf(jac *p) { memset(p, 0, sizeof *p); }
static int jac_is_inf(const jac *p) { return fe_is_zero(&p->Z); }

static void jac_neg(jac *r, const jac *p)
{
    fe zero = {{0, 0, 0, 0}};
    r->X = p->X; r->Z = p->Z;
    fe_sub(&r->Y, &zero, &p->Y);
}

/* dbl-2001-b (a = -3).  O and y = 0 give O, as ecp_j.c:258-263 */
static void jac_dbl(jac *r, const jac *p)
{
    fe delta, gamma, beta, alpha, t0, t1, X3, Y3, Z3;
    if (fe_is_zero(&p->Z) || fe_is_zero(&p->Y)) { jac_set_inf(r); return; }
    fe_sqr(&delta, &p->Z);
    fe_sqr(&gamma, &p->Y);
    fe_mul(&beta, &p->X, &gamma);
    fe_sub(&t0, &p->X, &delta);
    fe_add(&t1, &p->X, &delta);
    fe_mul(&alpha, &t0, &t1);
    fe_dbl(&t0, &alpha); fe_add(&alpha, &t0, &alpha);      /* 3 (X-d)(X+d) */
    fe_sqr(&X3, &alpha);
    fe_dbl(&t0, &beta); fe_dbl(&t0, &t0);                  /* 4 beta */
    fe_dbl(&t1, &t0);                                      /* 8 beta */
    fe_sub(&X3, &X3, &t1);
    fe_add(&Z3, &p->Y, &p->Z);
    fe_sqr(&Z3, &Z3);
    fe_sub(&Z3, &Z3, &gamma);
    fe_sub(&Z3, &Z3, &delta);
    fe_sub(&t0, &t0, &X3);
    fe_mul(&Y3, &alpha, &t0);
    fe_sqr(&t1, &gamma);
    fe_dbl(&t1, &t1); fe_dbl(&t1, &t1); fe_dbl(&t1, &t1);  /* 8 gamma^2 */
    fe_sub(&Y3, &Y3, &t1);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* add-2007-bl with the exceptional cases of ecp_j.c:416-427,455-464 */
static void jac_add(jac *r, const jac *a, const jac *b)
{
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    if (jac_is_inf(a)) { *r = *b; return; }
    if (jac_is_inf(b)) { *r = *a; return; }
    fe_sqr(&Z1Z1, &a->Z);
    fe_sqr(&Z2Z2, &b->Z);
    fe_mul(&U1, &a->X, &Z2Z2);
    fe_mul(&U2, &b->X, &Z1Z1);
    fe_mul(&S1, &b->Z, &Z2Z2); fe_mul(&S1, &a->Y, &S1);
    fe_mul(&S2, &a->Z, &Z1Z1); fe_mul(&S2, &b->Y, &S2);
    fe_sub(&H, &U2, &U1);
    if (fe_is_zero(&H)) {
        if (fe_eq(&S1, &S2)) { jac_dbl(r, a); } else { jac_set_inf(r); }
        return;
    }
    fe_dbl(&I, &H); fe_sqr(&I, &I);
    fe_mul(&J, &H, &I);
    fe_sub(&rr, &S2, &S1); fe_dbl(&rr, &rr);
    fe_mul(&V, &U1, &I);
    fe_sqr(&X3, &rr);
    fe_sub(&X3, &X3, &J);
    fe_dbl(&t, &V);
    fe_sub(&X3, &X3, &t);
    fe_sub(&t, &V, &X3);
    fe_mul(&Y3, &rr, &t);
    fe_mul(&t, &S1, &J); fe_dbl(&t, &t);
    fe_sub(&Y3, &Y3, &t);
    fe_add(&Z3, &a->Z, &b->Z);
    fe_sqr(&Z3, &Z3);
    fe_sub(&Z3, &Z3, &Z1Z1);
    fe_sub(&Z3, &Z3, &Z2Z2);
    fe_mul(&Z3, &Z3, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* returns 0 if p == O (ecp_j.c:119-121) */
static int jac_to_affine_x(fe *x, const jac *p)
{
    fe zi;
    if (jac_is_inf(p)) return 0;
    fe_inv(&zi, &p->Z);
    fe_sqr(&zi, &zi);
    fe_mul(x, &p->X, &zi);
    return 1;
}

/* ---------------------------------------------------------------- w-NAF --- */
#define NAF_W 5                                   /* ecNAFWidth(>=120 bits) = 5, ec.c:420-429 */
#define NAF_MAX 260

/* digits[i] in {0, +-1, +-3, .., +-15}; returns the length */
static int wnaf(int8_t digits[NAF_MAX], const uint64_t k_[5])
{
    uint64_t k[5];
    int len = 0;
    memcpy(k, k_, sizeof k);
    memset(digits, 0, NAF_MAX);
    while (k[0] | k[1] | k[2] | k[3] | k[4]) {
        int d = 0;
        if (k[0] & 1) {
            d = (int)(k[0] & ((1u << NAF_W) - 1));
            if (d >= (1 << (NAF_W - 1))) d -= (1 << NAF_W);
            /* k -= d */
            if (d > 0) {
                uint64_t borrow = (uint64_t)d;
                for (int i = 0; i < 5 && borrow; ++i) { uint64_t t = k[i]; k[i] = t - borrow; borrow = t < borrow; }
            } else {
                uint64_t carry = (uint64_t)(-d);
                for (int i = 0; i < 5 && carry; ++i) { k[i] += carry; carry = k[i] < carry; }
            }
        }
        digits[len++] = (int8_t)d;
        for (int i = 0; i < 4; ++i) k[i] = (k[i] >> 1) | (k[i + 1] << 63);
        k[4] >>= 1;
    }
    return len;
}

static void odd_multiples(jac tab[8], const jac *P)
{
    jac P2;
    jac_dbl(&P2, P);
    tab[0] = *P;
    for (int i = 1; i < 8; ++i) jac_add(&tab[i], &tab[i - 1], &P2);
}

/* R = u G + v Q (interleaved NAF, ec.c:1244-1268); returns 0 iff R == O */
static int double_mul_x(fe *rx, const uint64_t u[5], const jac *G, const uint64_t v[5], const jac *Q)
{
    int8_t du[NAF_MAX], dv[NAF_MAX];
    jac tg[8], tq[8], T, neg;
    int lu = wnaf(du, u), lv = wnaf(dv, v);
    int len = lu > lv ? lu : lv;
    odd_multiples(tg, G);
    odd_multiples(tq, Q);
    jac_set_inf(&T);
    for (int i = len - 1; i >= 0; --i) {
        jac_dbl(&T, &T);
        if (du[i] > 0) jac_add(&T, &T, &tg[du[i] >> 1]);
        else if (du[i] < 0) { jac_neg(&neg, &tg[(-du[i]) >> 1]); jac_add(&T, &T, &neg); }
        if (dv[i] > 0) jac_add(&T, &T, &tq[dv[i] >> 1]);
        else if (dv[i] < 0) { jac_neg(&neg, &tq[(-dv[i]) >> 1]); jac_add(&T, &T, &neg); }
    }
    return jac_to_affine_x(rx, &T);
}

/* ----------------------------------------------------------------- verify --- */
uint32_t orc_bign128Verify_ex(const uint8_t hash[32], const uint8_t sig[48],
                              const uint8_t pubkey[64], uint8_t rx_out[32])
{
    uint64_t q[4], s1[4], H[4], u[5], v[5];
    jac G, Q;
    fe rx;
    uint8_t msg[11 + 32 + 32], t[32];

    u256_from_le(q, Q_ORDER);
    /* Q: coordinates must be < p (qrFrom, bign_sign.c:306-311); no on-curve check */
    u256_from_le(Q.X.v, pubkey);
    u256_from_le(Q.Y.v, pubkey + 32);
    if (u256_cmp(Q.X.v, FE_P.v) >= 0 || u256_cmp(Q.Y.v, FE_P.v) >= 0) return ORC_BAD_PUBKEY;
    memset(&Q.Z, 0, sizeof Q.Z); Q.Z.v[0] = 1;
    /* s1 < q (:313-318) */
    u256_from_le(s1, sig + 16);
    if (u256_cmp(s1, q) >= 0) return ORC_BAD_SIG;
    /* s1 <- (s1 + H) mod q, H reduced by one conditional subtraction (:320-327) */
    u256_from_le(H, hash);
    if (u256_cmp(H, q) >= 0) u256_sub(H, H, q);
    {
        uint64_t carry = u256_add(u, s1, H);
        if (carry || u256_cmp(u, q) >= 0) u256_sub(u, u, q);
        u[4] = 0;
    }
    /* s0 + 2^l (:329-330) */
    memset(v, 0, sizeof v);
    for (int i = 0; i < 2; ++i) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; --k) w = (w << 8) | sig[8 * i + k];
        v[i] = w;
    }
    v[2] = 1;
    /* G = (0, yG) */
    memset(&G, 0, sizeof G);
    u256_from_le(G.Y.v, G_Y);
    G.Z.v[0] = 1;
    /* R = u G + v Q (:332) */
    if (!double_mul_x(&rx, u, &G, v, &Q)) return ORC_BAD_SIG;
    /* belt-hash(oid || <R.x> || H)[0..16) == s0 ? (:337-343) */
    memcpy(msg, OID_BELT_HASH, 11);
    u256_to_le(msg + 11, rx.v);
    memcpy(msg + 43, hash, 32);
    if (rx_out) memcpy(rx_out, msg + 11, 32);
    orc_beltHash(t, msg, sizeof msg);
    return memcmp(t, sig, 16) == 0 ? ORC_OK : ORC_BAD_SIG;
}

uint32_t orc_bign128Verify(const uint8_t hash[32], const uint8_t sig[48], const uint8_t pubkey[64])
{
    return orc_bign128Verify_ex(hash, sig, pubkey, 0);
}

typedef struct { const uint8_t *h, *s, *k; uint32_t *codes; } vjob;
static void verify_range(void *ctx, size_t lo, size_t hi)
{
    vjob *j = (vjob *)ctx;
    for (size_t i = lo; i < hi; ++i)
        j->codes[i] = orc_bign128Verify(j->h + 32 * i, j->s + 48 * i, j->k + 64 * i);
}
void orc_bign128Verify_batch(const uint8_t *hashes, const uint8_t *sigs, const uint8_t *pubkeys,
                             size_t n, uint32_t *codes, int nthreads)
{
    vjob j = {hashes, sigs, pubkeys, codes};
    (void)orc_beltH();
    orc_parallel_for(n, nthreads, verify_range, &j);
}

/* ---- reference driver (see bash_oracle.c) ---- */
typedef uint32_t (*ref_verify_fn)(const uint8_t *hash, const uint8_t *sig, const uint8_t *pubkey);
typedef struct { const uint8_t *h, *s, *k; uint32_t *codes; ref_verify_fn f; } ref_vjob;
static void ref_verify_range(void *ctx, size_t lo, size_t hi)
{
    ref_vjob *j = (ref_vjob *)ctx;
    for (size_t i = lo; i < hi; ++i) j->codes[i] = j->f(j->h + 32 * i, j->s + 48 * i, j->k + 64 * i);
}
void orc_drive_ref_verify(void *fn, const uint8_t *hashes, const uint8_t *sigs, const uint8_t *pubkeys,
                          size_t n, uint32_t *codes, int nthreads)
{
    ref_vjob j = {hashes, sigs, pubkeys, codes, (ref_verify_fn)fn};
    /* bign128Verify builds its process-global curve object on first use (bign128.c:34-88):
       take that hit once, single-threaded, before fanning out */
    if (n) codes[0] = j.f(hashes, sigs, pubkeys);
    orc_parallel_for(n, nthreads, ref_verify_range, &j);
}
