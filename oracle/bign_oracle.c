/*
 * bign_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of STB 34.101.45 signature verification on the three standard bign curves:
 * bign-curve256v1 (l = 128, bee2 module bign128), 384v1 (l = 192, bign192), 512v1 (l = 256,
 * bign256).
 *
 * Follows:
 *   bign128Verify   src/crypto/bign/bign128.c:151-153,177-185 (fixed belt-hash OID DER)
 *   bign192/256     src/crypto/bign/bign192.c:151-153,177-185, bign256.c:151-153,177-185
 *   bignVerifyEc    src/crypto/bign/bign_sign.c:268-347       (checks, s1+H mod q, s0+2^l, hash tail)
 *   ecAddMulA       src/math/ec.c:1183-1273   (interleaved width-5 NAF over the two scalars)
 *   ecPreSO         src/math/ec.c:164-196     (odd multiples 1,3,..,15)
 *   Jacobian ops    src/math/ecp/ecp_j.c:241-299 (dbl, a = -3), :397-497 (add), :104-133 (to affine)
 *                   -- exceptional cases (O operands, P = +-Q, y = 0) handled as there
 *   GF(p)           src/math/zm.c:214-263 (Crandall mul/sqr), src/math/zz/zz_red.c:71-105,
 *                   src/math/gfp.c:33-44 (inverse = a^(p-2))
 *   constants       src/crypto/bign/bign_params.c:34-178 (STB 34.101.45 annex B data)
 *
 * The point formulas are the textbook EFD ones (dbl-2001-b, add-2007-bl); the
 * Jacobian representative may differ from bee2's by a scalar factor, the affine
 * result -- the only thing the algorithm outputs -- cannot.
 */
#include "oracle.h"
#include "orc_threads.h"
#include <string.h>

typedef unsigned __int128 u128;
#define MAXW 8                                    /* 64-bit words of the largest field */
typedef struct { uint64_t v[MAXW]; } fe;           /* canonical residue in [0, p), words >= n are 0 */
typedef struct { fe X, Y, Z; } jac;                /* O  <=>  Z == 0 */

typedef struct {
    int n;                    /* 64-bit words per field element: 4, 6, 8 */
    uint64_t c;               /* p = 2^(64 n) - c */
    uint64_t p[MAXW], q[MAXW], yG[MAXW];
} curve;

/* STB 34.101.45 annex B, little-endian octets: group order q and y-coordinate of G = (0, yG) */
static const uint8_t Q128[32] = {
    0x07, 0x66, 0x3D, 0x26, 0x99, 0xBF, 0x5A, 0x7E, 0xFC, 0x4D, 0xFB, 0x0D, 0xD6, 0x8E, 0x5C, 0xD9,
    0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
static const uint8_t YG128[32] = {
    0x93, 0x6A, 0x51, 0x04, 0x18, 0xCF, 0x29, 0x1E, 0x52, 0xF6, 0x08, 0xC4, 0x66, 0x39, 0x91, 0x78,
    0x5D, 0x83, 0xD6, 0x51, 0xA3, 0xC9, 0xE4, 0x5C, 0x9F, 0xD6, 0x16, 0xFB, 0x3C, 0xFC, 0xF7, 0x6B};
static const uint8_t Q192[48] = {
    0xB7, 0xA7, 0x0C, 0xF3, 0x3F, 0xDC, 0xB7, 0x3D, 0x0A, 0xFF, 0xA4, 0xA6, 0xE7, 0xDA, 0x46, 0x80,
    0xBB, 0x7B, 0xAF, 0x73, 0x03, 0xC4, 0xCC, 0x6C, 0xFE, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF,
    0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
static const uint8_t YG192[48] = {
    0x51, 0xC4, 0x33, 0xF7, 0x31, 0xCB, 0x5E, 0xEA, 0xF9, 0x42, 0x2A, 0x6B, 0x27, 0x3E, 0x40, 0x84,
    0x55, 0xD3, 0xB1, 0x66, 0x9E, 0xE7, 0x49, 0x05, 0xA0, 0xFF, 0x86, 0xDC, 0x11, 0x9A, 0x72, 0x3A,
    0x89, 0xBF, 0x2D, 0x43, 0x7E, 0x11, 0x30, 0x63, 0x9E, 0x9E, 0x2E, 0xA8, 0x24, 0x82, 0x43, 0x5D};
static const uint8_t Q256[64] = {
    0xF1, 0x8E, 0x06, 0x0D, 0x49, 0xAD, 0xFF, 0xDC, 0x32, 0xDF, 0x56, 0x95, 0xE5, 0xCA, 0x1B, 0x36,
    0xF4, 0x13, 0x21, 0x2E, 0xB0, 0xEB, 0x6B, 0xF2, 0x4E, 0x00, 0x98, 0x01, 0x2C, 0x09, 0xC0, 0xB2,
    0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF,
    0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
static const uint8_t YG256[64] = {
    0xBD, 0xED, 0xEF, 0xCE, 0x6F, 0xAE, 0x92, 0xB7, 0x04, 0x0D, 0x4C, 0xC9, 0xB9, 0x83, 0xAA, 0x67,
    0x61, 0x22, 0xE8, 0xEE, 0x95, 0x73, 0x77, 0xFF, 0xD2, 0x6F, 0xFA, 0x0E, 0xE2, 0xDD, 0x73, 0x69,
    0xDA, 0xCA, 0xCC, 0x00, 0x1B, 0xF8, 0xED, 0xD2, 0xE2, 0xBC, 0x61, 0xB3, 0xB3, 0x41, 0xAB, 0xB0,
    0xAB, 0x8F, 0xD1, 0xA0, 0xF7, 0xE6, 0x82, 0xB1, 0x81, 0x76, 0x03, 0xE4, 0x7A, 0xFF, 0x26, 0xA8};

/* curve coefficient b (a = p - 3 on all three curves), bign_params.c:43-56,87-104,142-165 */
static const uint8_t B128[32] = {
    0xF1, 0x03, 0x9C, 0xD6, 0x6B, 0x7D, 0x2E, 0xB2, 0x53, 0x92, 0x8B, 0x97, 0x69, 0x50, 0xF5, 0x4C,
    0xBE, 0xFB, 0xD8, 0xE4, 0xAB, 0x3A, 0xC1, 0xD2, 0xED, 0xA8, 0xF3, 0x15, 0x15, 0x6C, 0xCE, 0x77};
static const uint8_t B192[48] = {
    0x64, 0xBF, 0x73, 0x68, 0x23, 0xFC, 0xA7, 0xBC, 0x7C, 0xBD, 0xCE, 0xF3, 0xF0, 0xE2, 0xBD, 0x14,
    0x3A, 0x2E, 0x71, 0xE9, 0xF9, 0x6A, 0x21, 0xA6, 0x96, 0xB1, 0xFB, 0x0F, 0xBB, 0x48, 0x27, 0x71,
    0xD2, 0x34, 0x5D, 0x65, 0xAB, 0x5A, 0x07, 0x33, 0x20, 0xEF, 0x9C, 0x95, 0xE1, 0xDF, 0x75, 0x3C};
static const uint8_t B256[64] = {
    0x90, 0x9C, 0x13, 0xD6, 0x98, 0x69, 0x34, 0x09, 0x7A, 0xA2, 0x49, 0x3A, 0x27, 0x22, 0x86, 0xEA,
    0x43, 0xA2, 0xAC, 0x87, 0x8C, 0x00, 0x33, 0x29, 0x95, 0x5E, 0x24, 0xC4, 0xB5, 0xDC, 0x11, 0x27,
    0x88, 0xB0, 0xAD, 0xDA, 0xE3, 0x13, 0xCE, 0x17, 0x51, 0x25, 0x5D, 0xDD, 0xEE, 0xA9, 0xC6, 0x5B,
    0x89, 0x58, 0xFD, 0x60, 0x6A, 0x5D, 0x8C, 0xD8, 0x43, 0x8C, 0x3B, 0x93, 0x44, 0x59, 0xB4, 0x6C};

/* longest DER OID the checker takes (stack buffers below); the reference takes any length */
#define ORC_OID_MAX 8192
/* DER of the pre-hash OIDs of the level-fixed facades: belt-hash, bash384, bash512 */
static const uint8_t OID_BELT_HASH[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};
static const uint8_t OID_BASH384[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x4D, 0x0C};
static const uint8_t OID_BASH512[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x4D, 0x0D};

/* ------------------------------------------------------------- integers --- */
static void words_from_le(uint64_t *w, const uint8_t *p, int n)
{
    for (int i = 0; i < n; ++i) {
        uint64_t v = 0;
        for (int k = 7; k >= 0; --k) v = (v << 8) | p[8 * i + k];
        w[i] = v;
    }
}
static void words_to_le(uint8_t *p, const uint64_t *w, int n)
{
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 8; ++k) p[8 * i + k] = (uint8_t)(w[i] >> (8 * k));
}
static int words_cmp(const uint64_t *a, const uint64_t *b, int n)
{
    for (int i = n - 1; i >= 0; --i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
static uint64_t words_add(uint64_t *r, const uint64_t *a, const uint64_t *b, int n)
{
    u128 c = 0;
    for (int i = 0; i < n; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t words_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n)
{
    uint64_t borrow = 0;
    for (int i = 0; i < n; ++i) {
        uint64_t t = a[i] - b[i];
        uint64_t b2 = (a[i] < b[i]) | ((t < borrow) ? 1u : 0u);
        r[i] = t - borrow;
        borrow = b2;
    }
    return borrow;
}

static void curve_init(curve *E, int l)
{
    memset(E, 0, sizeof *E);
    E->n = l / 32;
    E->c = l == 128 ? 189 : l == 192 ? 317 : 569;
    for (int i = 0; i < E->n; ++i) E->p[i] = ~0ull;
    E->p[0] = 0ull - E->c;
    words_from_le(E->q, l == 128 ? Q128 : l == 192 ? Q192 : Q256, E->n);
    words_from_le(E->yG, l == 128 ? YG128 : l == 192 ? YG192 : YG256, E->n);
}

/* ----------------------------------------------------------------- GF(p) --- */
static int fe_is_zero(const curve *E, const fe *a)
{
    uint64_t z = 0;
    for (int i = 0; i < E->n; ++i) z |= a->v[i];
    return z == 0;
}
static int fe_eq(const curve *E, const fe *a, const fe *b) { return words_cmp(a->v, b->v, E->n) == 0; }

static void fe_add(const curve *E, fe *r, const fe *a, const fe *b)
{
    uint64_t t[MAXW] = {0}, carry = words_add(t, a->v, b->v, E->n);
    if (carry || words_cmp(t, E->p, E->n) >= 0) words_sub(t, t, E->p, E->n);
    memcpy(r->v, t, sizeof t);
}
static void fe_sub(const curve *E, fe *r, const fe *a, const fe *b)
{
    uint64_t t[MAXW] = {0};
    if (words_sub(t, a->v, b->v, E->n)) words_add(t, t, E->p, E->n);
    memcpy(r->v, t, sizeof t);
}
static void fe_dbl(const curve *E, fe *r, const fe *a) { fe_add(E, r, a, a); }

/* schoolbook (zz_mul.c:82-105) then Crandall fold by c (zz_red.c:71-105) */
static void fe_mul(const curve *E, fe *r, const fe *a, const fe *b)
{
    const int n = E->n;
    uint64_t w[2 * MAXW] = {0};
    for (int i = 0; i < n; ++i) {
        u128 carry = 0;
        for (int j = 0; j < n; ++j) {
            carry += (u128)a->v[i] * b->v[j] + w[i + j];
            w[i + j] = (uint64_t)carry;
            carry >>= 64;
        }
        w[i + n] = (uint64_t)carry;
    }
    /* lo + c * hi  ->  n + 1 words */
    uint64_t t[MAXW] = {0};
    u128 acc = 0;
    for (int i = 0; i < n; ++i) {
        acc += (u128)w[n + i] * E->c + w[i];
        t[i] = (uint64_t)acc;
        acc >>= 64;
    }
    /* top word (< c + 1) folds once more; a final carry folds as +c */
    acc = (u128)(uint64_t)acc * E->c;
    for (int i = 0; i < n; ++i) { acc += t[i]; t[i] = (uint64_t)acc; acc >>= 64; }
    if (acc) {
        acc = E->c;
        for (int i = 0; i < n; ++i) { acc += t[i]; t[i] = (uint64_t)acc; acc >>= 64; }
    }
    if (words_cmp(t, E->p, n) >= 0) words_sub(t, t, E->p, n);
    memcpy(r->v, t, sizeof t);
}
static void fe_sqr(const curve *E, fe *r, const fe *a) { fe_mul(E, r, a, a); }

/* a^(p-2) by plain left-to-right square-and-multiply (gfp.c:33-44: Fermat inverse) */
static void fe_inv(const curve *E, fe *r, const fe *a)
{
    fe x = *a, acc;
    uint64_t e[MAXW];
    memcpy(e, E->p, sizeof e);
    e[0] -= 2;
    memset(&acc, 0, sizeof acc);
    acc.v[0] = 1;
    for (int i = 64 * E->n - 1; i >= 0; --i) {
        fe_sqr(E, &acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(E, &acc, &acc, &x);
    }
    *r = acc;
}

/* -------------------------------------------------------------- Jacobian --- */
static void jac_set_inf(jac *p) { memset(p, 0, sizeof *p); }
static int jac_is_inf(const curve *E, const jac *p) { return fe_is_zero(E, &p->Z); }

static void jac_neg(const curve *E, jac *r, const jac *p)
{
    fe zero;
    memset(&zero, 0, sizeof zero);
    r->X = p->X; r->Z = p->Z;
    fe_sub(E, &r->Y, &zero, &p->Y);
}

/* dbl-2001-b (a = -3).  O and y = 0 give O, as ecp_j.c:258-263 */
static void jac_dbl(const curve *E, jac *r, const jac *p)
{
    fe delta, gamma, beta, alpha, t0, t1, X3, Y3, Z3;
    if (fe_is_zero(E, &p->Z) || fe_is_zero(E, &p->Y)) { jac_set_inf(r); return; }
    fe_sqr(E, &delta, &p->Z);
    fe_sqr(E, &gamma, &p->Y);
    fe_mul(E, &beta, &p->X, &gamma);
    fe_sub(E, &t0, &p->X, &delta);
    fe_add(E, &t1, &p->X, &delta);
    fe_mul(E, &alpha, &t0, &t1);
    fe_dbl(E, &t0, &alpha); fe_add(E, &alpha, &t0, &alpha);      /* 3 (X-d)(X+d) */
    fe_sqr(E, &X3, &alpha);
    fe_dbl(E, &t0, &beta); fe_dbl(E, &t0, &t0);                  /* 4 beta */
    fe_dbl(E, &t1, &t0);                                         /* 8 beta */
    fe_sub(E, &X3, &X3, &t1);
    fe_add(E, &Z3, &p->Y, &p->Z);
    fe_sqr(E, &Z3, &Z3);
    fe_sub(E, &Z3, &Z3, &gamma);
    fe_sub(E, &Z3, &Z3, &delta);
    fe_sub(E, &t0, &t0, &X3);
    fe_mul(E, &Y3, &alpha, &t0);
    fe_sqr(E, &t1, &gamma);
    fe_dbl(E, &t1, &t1); fe_dbl(E, &t1, &t1); fe_dbl(E, &t1, &t1);  /* 8 gamma^2 */
    fe_sub(E, &Y3, &Y3, &t1);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* add-2007-bl with the exceptional cases of ecp_j.c:416-427,455-464 */
static void jac_add(const curve *E, jac *r, const jac *a, const jac *b)
{
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    if (jac_is_inf(E, a)) { *r = *b; return; }
    if (jac_is_inf(E, b)) { *r = *a; return; }
    fe_sqr(E, &Z1Z1, &a->Z);
    fe_sqr(E, &Z2Z2, &b->Z);
    fe_mul(E, &U1, &a->X, &Z2Z2);
    fe_mul(E, &U2, &b->X, &Z1Z1);
    fe_mul(E, &S1, &b->Z, &Z2Z2); fe_mul(E, &S1, &a->Y, &S1);
    fe_mul(E, &S2, &a->Z, &Z1Z1); fe_mul(E, &S2, &b->Y, &S2);
    fe_sub(E, &H, &U2, &U1);
    if (fe_is_zero(E, &H)) {
        if (fe_eq(E, &S1, &S2)) { jac_dbl(E, r, a); } else { jac_set_inf(r); }
        return;
    }
    fe_dbl(E, &I, &H); fe_sqr(E, &I, &I);
    fe_mul(E, &J, &H, &I);
    fe_sub(E, &rr, &S2, &S1); fe_dbl(E, &rr, &rr);
    fe_mul(E, &V, &U1, &I);
    fe_sqr(E, &X3, &rr);
    fe_sub(E, &X3, &X3, &J);
    fe_dbl(E, &t, &V);
    fe_sub(E, &X3, &X3, &t);
    fe_sub(E, &t, &V, &X3);
    fe_mul(E, &Y3, &rr, &t);
    fe_mul(E, &t, &S1, &J); fe_dbl(E, &t, &t);
    fe_sub(E, &Y3, &Y3, &t);
    fe_add(E, &Z3, &a->Z, &b->Z);
    fe_sqr(E, &Z3, &Z3);
    fe_sub(E, &Z3, &Z3, &Z1Z1);
    fe_sub(E, &Z3, &Z3, &Z2Z2);
    fe_mul(E, &Z3, &Z3, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* returns 0 if p == O (ecp_j.c:119-121) */
static int jac_to_affine_x(const curve *E, fe *x, const jac *p)
{
    fe zi;
    if (jac_is_inf(E, p)) return 0;
    fe_inv(E, &zi, &p->Z);
    fe_sqr(E, &zi, &zi);
    fe_mul(E, x, &p->X, &zi);
    return 1;
}

/* ---------------------------------------------------------------- w-NAF --- */
#define NAF_W 5                                   /* ecNAFWidth(120..335 bits) = 5, ec.c:420-429 */
#define NAF_MAX 520
#define SCW (MAXW + 1)

/* digits[i] in {0, +-1, +-3, .., +-15}; returns the length.  (For the 512-bit scalar the
   reference uses width 6; the multiple computed is the same.) */
static int wnaf(int8_t digits[NAF_MAX], const uint64_t k_[SCW])
{
    uint64_t k[SCW];
    int len = 0;
    memcpy(k, k_, sizeof k);
    memset(digits, 0, NAF_MAX);
    for (;;) {
        uint64_t any = 0;
        for (int i = 0; i < SCW; ++i) any |= k[i];
        if (!any) break;
        int d = 0;
        if (k[0] & 1) {
            d = (int)(k[0] & ((1u << NAF_W) - 1));
            if (d >= (1 << (NAF_W - 1))) d -= (1 << NAF_W);
            if (d > 0) {
                uint64_t borrow = (uint64_t)d;
                for (int i = 0; i < SCW && borrow; ++i) { uint64_t t = k[i]; k[i] = t - borrow; borrow = t < borrow; }
            } else {
                uint64_t carry = (uint64_t)(-d);
                for (int i = 0; i < SCW && carry; ++i) { k[i] += carry; carry = k[i] < carry; }
            }
        }
        digits[len++] = (int8_t)d;
        for (int i = 0; i < SCW - 1; ++i) k[i] = (k[i] >> 1) | (k[i + 1] << 63);
        k[SCW - 1] >>= 1;
    }
    return len;
}

static void odd_multiples(const curve *E, jac tab[8], const jac *P)
{
    jac P2;
    jac_dbl(E, &P2, P);
    tab[0] = *P;
    for (int i = 1; i < 8; ++i) jac_add(E, &tab[i], &tab[i - 1], &P2);
}

/* R = u G + v Q (interleaved NAF, ec.c:1244-1268); returns 0 iff R == O */
static int double_mul_x(const curve *E, fe *rx, const uint64_t u[SCW], const jac *G, const uint64_t v[SCW], const jac *Q)
{
    int8_t du[NAF_MAX], dv[NAF_MAX];
    jac tg[8], tq[8], T, neg;
    int lu = wnaf(du, u), lv = wnaf(dv, v);
    int len = lu > lv ? lu : lv;
    odd_multiples(E, tg, G);
    odd_multiples(E, tq, Q);
    jac_set_inf(&T);
    for (int i = len - 1; i >= 0; --i) {
        jac_dbl(E, &T, &T);
        if (du[i] > 0) jac_add(E, &T, &T, &tg[du[i] >> 1]);
        else if (du[i] < 0) { jac_neg(E, &neg, &tg[(-du[i]) >> 1]); jac_add(E, &T, &T, &neg); }
        if (dv[i] > 0) jac_add(E, &T, &T, &tq[dv[i] >> 1]);
        else if (dv[i] < 0) { jac_neg(E, &neg, &tq[(-dv[i]) >> 1]); jac_add(E, &T, &T, &neg); }
    }
    return jac_to_affine_x(E, rx, &T);
}

/* ----------------------------------------------------------------- verify --- */
/* bignVerify(params(l), oid_der, oid_len, hash[l/4], sig[3l/8], pubkey[l/2]) */
uint32_t orc_bignVerify_ex(size_t l, const uint8_t *oid_der, size_t oid_len, const uint8_t *hash,
                           const uint8_t *sig, const uint8_t *pubkey, uint8_t *rx_out)
{
    curve E;
    uint64_t s1[MAXW] = {0}, H[MAXW] = {0}, u[SCW] = {0}, v[SCW] = {0};
    jac G, Q;
    fe rx;
    uint8_t msg[ORC_OID_MAX + 128], t[32];
    if (l != 128 && l != 192 && l != 256) return ORC_BAD_PARAMS;
    if (oid_len > ORC_OID_MAX) return ORC_BAD_OID;                /* checker limit, not the reference's */
    curve_init(&E, (int)l);
    const int n = E.n, no = (int)l / 4;
    memset(&Q, 0, sizeof Q);
    memset(&G, 0, sizeof G);
    /* Q: coordinates must be < p (qrFrom, bign_sign.c:306-311); no on-curve check */
    words_from_le(Q.X.v, pubkey, n);
    words_from_le(Q.Y.v, pubkey + no, n);
    if (words_cmp(Q.X.v, E.p, n) >= 0 || words_cmp(Q.Y.v, E.p, n) >= 0) return ORC_BAD_PUBKEY;
    Q.Z.v[0] = 1;
    /* s1 < q (:313-318) */
    words_from_le(s1, sig + no / 2, n);
    if (words_cmp(s1, E.q, n) >= 0) return ORC_BAD_SIG;
    /* s1 <- (s1 + H) mod q, H reduced by one conditional subtraction (:320-327) */
    words_from_le(H, hash, n);
    if (words_cmp(H, E.q, n) >= 0) words_sub(H, H, E.q, n);
    {
        uint64_t carry = words_add(u, s1, H, n);
        if (carry || words_cmp(u, E.q, n) >= 0) words_sub(u, u, E.q, n);
    }
    /* s0 + 2^l (:329-330) */
    for (int i = 0; i < n / 2; ++i) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; --k) w = (w << 8) | sig[8 * i + k];
        v[i] = w;
    }
    v[n / 2] = 1;
    /* G = (0, yG) */
    memcpy(G.Y.v, E.yG, sizeof E.yG);
    G.Z.v[0] = 1;
    /* R = u G + v Q (:332) */
    if (!double_mul_x(&E, &rx, u, &G, v, &Q)) return ORC_BAD_SIG;
    /* belt-hash(oid || <R.x> || H)[0..l/8) == s0 ? (:337-343) */
    memcpy(msg, oid_der, oid_len);
    words_to_le(msg + oid_len, rx.v, n);
    memcpy(msg + oid_len + no, hash, no);
    if (rx_out) memcpy(rx_out, msg + oid_len, no);
    orc_beltHash(t, msg, oid_len + 2 * no);
    return memcmp(t, sig, no / 2) == 0 ? ORC_OK : ORC_BAD_SIG;
}

/* bignPubkeyVal: src/crypto/bign/bign_misc.c:319-365 -- both coordinates < p (qrFrom), then
   ecpIsOnA, src/math/ecp/ecp_a.c:36-60: (x^2 + a) x + b == y^2 */
uint32_t orc_bignPubkeyVal(size_t l, const uint8_t *pubkey)
{
    curve E;
    fe x, y, t, a, b;
    if (l != 128 && l != 192 && l != 256) return ORC_BAD_PARAMS;
    curve_init(&E, (int)l);
    const int n = E.n, no = (int)l / 4;
    memset(&x, 0, sizeof x); memset(&y, 0, sizeof y); memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
    words_from_le(x.v, pubkey, n);
    words_from_le(y.v, pubkey + no, n);
    if (words_cmp(x.v, E.p, n) >= 0 || words_cmp(y.v, E.p, n) >= 0) return ORC_BAD_PUBKEY;
    memcpy(a.v, E.p, sizeof E.p);
    a.v[0] -= 3;                                        /* a = p - 3, no borrow: p[0] = -c */
    words_from_le(b.v, l == 128 ? B128 : l == 192 ? B192 : B256, n);
    fe_sqr(&E, &t, &x);
    fe_add(&E, &t, &t, &a);
    fe_mul(&E, &t, &t, &x);
    fe_add(&E, &t, &t, &b);
    fe_sqr(&E, &y, &y);
    return fe_eq(&E, &t, &y) ? ORC_OK : ORC_BAD_PUBKEY;
}
void orc_bignPubkeyVal_batch(size_t l, const uint8_t *pubkeys, size_t n, uint32_t *codes)
{
    for (size_t i = 0; i < n; ++i) codes[i] = orc_bignPubkeyVal(l, pubkeys + i * (l / 2));
}

uint32_t orc_bign128Verify_ex(const uint8_t hash[32], const uint8_t sig[48], const uint8_t pubkey[64], uint8_t rx[32])
{
    return orc_bignVerify_ex(128, OID_BELT_HASH, 11, hash, sig, pubkey, rx);
}
uint32_t orc_bign128Verify(const uint8_t hash[32], const uint8_t sig[48], const uint8_t pubkey[64])
{
    return orc_bignVerify_ex(128, OID_BELT_HASH, 11, hash, sig, pubkey, 0);
}
uint32_t orc_bign192Verify(const uint8_t hash[48], const uint8_t sig[72], const uint8_t pubkey[96])
{
    return orc_bignVerify_ex(192, OID_BASH384, 11, hash, sig, pubkey, 0);
}
uint32_t orc_bign256Verify(const uint8_t hash[64], const uint8_t sig[96], const uint8_t pubkey[128])
{
    return orc_bignVerify_ex(256, OID_BASH512, 11, hash, sig, pubkey, 0);
}

typedef struct { size_t l; const uint8_t *oid; size_t oid_len; const uint8_t *h, *s, *k; uint32_t *codes; } vjob;
static void verify_range(void *ctx, size_t lo, size_t hi)
{
    vjob *j = (vjob *)ctx;
    const size_t no = j->l / 4;
    for (size_t i = lo; i < hi; ++i)
        j->codes[i] = orc_bignVerify_ex(j->l, j->oid, j->oid_len, j->h + no * i, j->s + (no + no / 2) * i,
                                        j->k + 2 * no * i, 0);
}
void orc_bignVerify_batch(size_t l, const uint8_t *oid_der, size_t oid_len, const uint8_t *hashes,
                          const uint8_t *sigs, const uint8_t *pubkeys, size_t n, uint32_t *codes, int nthreads)
{
    vjob j = {l, oid_der, oid_len, hashes, sigs, pubkeys, codes};
    (void)orc_beltH();
    orc_parallel_for(n, nthreads, verify_range, &j);
}
void orc_bign128Verify_batch(const uint8_t *hashes, const uint8_t *sigs, const uint8_t *pubkeys,
                             size_t n, uint32_t *codes, int nthreads)
{
    orc_bignVerify_batch(128, OID_BELT_HASH, 11, hashes, sigs, pubkeys, n, codes, nthreads);
}

/* ============================================ SURVEY.md 8f-4, second half ===
 * Key generation, public key from private key and signing on the three standard curves.
 * Follows
 *   bignPubkeyCalcEc   src/crypto/bign/bign_misc.c:373-417   (0 < d < q, Q = d G)
 *   bignKeypairGenEc   src/crypto/bign/bign_misc.c:182-229   (d from zzRandNZMod, Q = d G)
 *   zzRandNZMod        src/math/zz/zz_mod.c:463-485          (rng -> l bits, retry while 0 or >= mod)
 *   bignSignEc         src/crypto/bign/bign_sign.c:32-112    (k from rng)
 *   bignSign2Ec        src/crypto/bign/bign_sign.c:140-245   (k by STB 34.101.45 alg. 6.3.3)
 *   zzSubMod           src/math/zz/zz_mod.c:120-132          (a - b, + mod on borrow; b is NOT reduced first)
 * The scalar multiplication is a plain left-to-right double-and-add on the Jacobian code above
 * (the reference uses precomputed tables and regular recoding, bignMulBase; the affine result is
 * the same point).  Nothing here is constant-time -- this is the checker.
 */
#define ORC_BAD_PRIVKEY 504u
#define ORC_BAD_RNG     304u

/* R = k G, affine (x, y); 0 iff R == O */
static int mul_base(const curve *E, fe *x, fe *y, const uint64_t k[MAXW])
{
    jac G, T;
    fe zi, zi2;
    memset(&G, 0, sizeof G);
    memcpy(G.Y.v, E->yG, sizeof E->yG);
    G.Z.v[0] = 1;
    jac_set_inf(&T);
    for (int i = 64 * E->n - 1; i >= 0; --i) {
        jac_dbl(E, &T, &T);
        if ((k[i >> 6] >> (i & 63)) & 1) jac_add(E, &T, &T, &G);
    }
    if (jac_is_inf(E, &T)) return 0;
    fe_inv(E, &zi, &T.Z);
    fe_sqr(E, &zi2, &zi);
    fe_mul(E, x, &T.X, &zi2);
    fe_mul(E, &zi2, &zi2, &zi);
    fe_mul(E, y, &T.Y, &zi2);
    return 1;
}
static int words_is_zero(const uint64_t *a, int n)
{
    uint64_t z = 0;
    for (int i = 0; i < n; ++i) z |= a[i];
    return z == 0;
}
/* x[0..nx) mod q by binary long division (test code: clarity over speed) */
static void mod_q(const curve *E, uint64_t r[MAXW], const uint64_t *x, int nx)
{
    const int n = E->n;
    uint64_t t[MAXW + 1];
    memset(t, 0, sizeof t);
    for (int bit = 64 * nx - 1; bit >= 0; --bit) {
        for (int i = n; i > 0; --i) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
        t[0] = (t[0] << 1) | ((x[bit >> 6] >> (bit & 63)) & 1);
        if (t[n] || words_cmp(t, E->q, n) >= 0) { uint64_t b = words_sub(t, t, E->q, n); t[n] -= b; }
    }
    memset(r, 0, sizeof(uint64_t) * MAXW);
    memcpy(r, t, sizeof(uint64_t) * n);
}
/* zzSubMod semantics: c = a - b (mod 2^(64 n)), + q if the subtraction borrowed */
static void sub_mod_q(const curve *E, uint64_t *c, const uint64_t *a, const uint64_t *b)
{
    if (words_sub(c, a, b, E->n)) words_add(c, c, E->q, E->n);
}

uint32_t orc_bignPubkeyCalc(size_t l, uint8_t *pubkey, const uint8_t *privkey)
{
    curve E;
    uint64_t d[MAXW] = {0};
    fe x, y;
    if (l != 128 && l != 192 && l != 256) return ORC_BAD_PARAMS;
    curve_init(&E, (int)l);
    words_from_le(d, privkey, E.n);
    if (words_is_zero(d, E.n) || words_cmp(d, E.q, E.n) >= 0) return ORC_BAD_PRIVKEY;
    if (!mul_base(&E, &x, &y, d)) return ORC_BAD_PARAMS;
    words_to_le(pubkey, x.v, E.n);
    words_to_le(pubkey + l / 4, y.v, E.n);
    return ORC_OK;
}

/* bignKeypairGen with the rng replaced by its output stream: `rnd` holds consecutive l/4-octet draws
   (what gen_i would have written), *used tells how many were consumed.  NOTE: the reference draws d
   below ec->f->mod = p (bign_misc.c:209), not below q. */
uint32_t orc_bignKeypairGen(size_t l, uint8_t *privkey, uint8_t *pubkey, const uint8_t *rnd, size_t ndraws, size_t *used)
{
    curve E;
    uint64_t d[MAXW] = {0};
    size_t i = 0;
    if (l != 128 && l != 192 && l != 256) return ORC_BAD_PARAMS;
    curve_init(&E, (int)l);
    for (;; ++i) {
        if (i == ndraws) return ORC_BAD_RNG;
        words_from_le(d, rnd + i * (l / 4), E.n);        /* wwBitSize(p) = 2l: nothing to trim */
        if (!words_is_zero(d, E.n) && words_cmp(d, E.p, E.n) < 0) break;
    }
    if (used) *used = i + 1;
    words_to_le(privkey, d, E.n);
    {
        fe x, y;
        if (!mul_base(&E, &x, &y, d)) return ORC_BAD_PARAMS;
        words_to_le(pubkey, x.v, E.n);
        words_to_le(pubkey + l / 4, y.v, E.n);
    }
    return ORC_OK;
}

/* the common tail of bignSignEc / bignSign2Ec from "R <- k G" on (bign_sign.c:84-108 = :214-241) */
static uint32_t sign_with_k(const curve *E, size_t l, uint8_t *sig, const uint8_t *oid_der, size_t oid_len,
                            const uint8_t *hash, const uint64_t d[MAXW], const uint64_t k[MAXW])
{
    const int n = E->n, no = (int)l / 4;
    fe x, y;
    uint8_t msg[ORC_OID_MAX + 128], t[32];
    uint64_t s0[MAXW] = {0}, prod[2 * MAXW] = {0}, s1[MAXW], H[MAXW] = {0};
    if (!mul_base(E, &x, &y, k)) return ORC_BAD_PARAMS;
    memcpy(msg, oid_der, oid_len);
    words_to_le(msg + oid_len, x.v, n);
    memcpy(msg + oid_len + no, hash, no);
    orc_beltHash(t, msg, oid_len + 2 * no);
    memcpy(sig, t, no / 2);                                      /* s0 = first l bits */
    /* (s0 + 2^l) d */
    for (int i = 0; i < n / 2; ++i) {
        uint64_t w = 0;
        for (int b = 7; b >= 0; --b) w = (w << 8) | sig[8 * i + b];
        s0[i] = w;
    }
    s0[n / 2] = 1;
    for (int i = 0; i <= n / 2; ++i) {
        u128 carry = 0;
        for (int j = 0; j < n; ++j) {
            carry += (u128)s0[i] * d[j] + prod[i + j];
            prod[i + j] = (uint64_t)carry;
            carry >>= 64;
        }
        prod[i + n] += (uint64_t)carry;
    }
    mod_q(E, s1, prod, n + n / 2 + 1);
    sub_mod_q(E, s1, k, s1);                                     /* k - (s0 + 2^l) d */
    words_from_le(H, hash, n);
    sub_mod_q(E, s1, s1, H);                                     /* ... - H, H as it is */
    words_to_le(sig + no / 2, s1, n);
    return ORC_OK;
}

/* OID DER syntax is not re-checked here (as in orc_bignVerify_ex): the product's check is pinned on its own
   (tests/golden/bign_oid_der.json) */
/* bignSign with the rng replaced by its output: k = first draw in {1..q-1} */
uint32_t orc_bignSign_rnd(size_t l, uint8_t *sig, const uint8_t *oid_der, size_t oid_len, const uint8_t *hash,
                          const uint8_t *privkey, const uint8_t *rnd, size_t ndraws, size_t *used)
{
    curve E;
    uint64_t d[MAXW] = {0}, k[MAXW] = {0};
    size_t i = 0;
    if (l != 128 && l != 192 && l != 256) return ORC_BAD_PARAMS;
    if (oid_len > ORC_OID_MAX) return ORC_BAD_OID;                /* checker limit, not the reference's */
    curve_init(&E, (int)l);
    words_from_le(d, privkey, E.n);
    if (words_is_zero(d, E.n) || words_cmp(d, E.q, E.n) >= 0) return ORC_BAD_PRIVKEY;
    for (;; ++i) {
        if (i == ndraws) return ORC_BAD_RNG;
        words_from_le(k, rnd + i * (l / 4), E.n);
        if (!words_is_zero(k, E.n) && words_cmp(k, E.q, E.n) < 0) break;
    }
    if (used) *used = i + 1;
    return sign_with_k(&E, l, sig, oid_der, oid_len, hash, d, k);
}

uint32_t orc_bignSign2(size_t l, uint8_t *sig, const uint8_t *oid_der, size_t oid_len, const uint8_t *hash,
                       const uint8_t *privkey, const void *t, size_t t_len)
{
    curve E;
    uint64_t d[MAXW] = {0}, k[MAXW] = {0};
    uint8_t theta[32], kb[64], *msg;
    uint32_t K[8];
    if (l != 128 && l != 192 && l != 256) return ORC_BAD_PARAMS;
    if (oid_len > ORC_OID_MAX) return ORC_BAD_OID;                /* checker limit, not the reference's */
    curve_init(&E, (int)l);
    const int n = E.n, no = (int)l / 4;
    words_from_le(d, privkey, n);
    if (words_is_zero(d, n) || words_cmp(d, E.q, n) >= 0) return ORC_BAD_PRIVKEY;
    /* theta = belt-hash(oid || d || t) (:197-202) */
    {
        uint8_t stack_msg[ORC_OID_MAX + 64 + 256];
        msg = stack_msg;
        if (t_len > 256) return ORC_BAD_INPUT;                     /* checker limit, not the reference's */
        memcpy(msg, oid_der, oid_len);
        memcpy(msg + oid_len, privkey, no);
        if (t) memcpy(msg + oid_len + no, t, t_len);
        orc_beltHash(theta, msg, oid_len + no + (t ? t_len : 0));
    }
    /* k = H; k <- belt-wbl_theta(k) until k in {1..q-1} (:203-216) */
    orc_beltKeyExpand2(K, theta, 32);
    memcpy(kb, hash, no);
    for (;;) {
        orc_beltWBL(kb, (size_t)no / 16, K, 0);
        words_from_le(k, kb, n);
        if (!words_is_zero(k, n) && words_cmp(k, E.q, n) < 0) break;
    }
    return sign_with_k(&E, l, sig, oid_der, oid_len, hash, d, k);
}

/* ---- reference driver (see bash_oracle.c) ---- */
typedef uint32_t (*ref_verify_fn)(const uint8_t *hash, const uint8_t *sig, const uint8_t *pubkey);
typedef struct { const uint8_t *h, *s, *k; uint32_t *codes; ref_verify_fn f; size_t no; } ref_vjob;
static void ref_verify_range(void *ctx, size_t lo, size_t hi)
{
    ref_vjob *j = (ref_vjob *)ctx;
    for (size_t i = lo; i < hi; ++i)
        j->codes[i] = j->f(j->h + j->no * i, j->s + (j->no + j->no / 2) * i, j->k + 2 * j->no * i);
}
void orc_drive_ref_verify(void *fn, const uint8_t *hashes, const uint8_t *sigs, const uint8_t *pubkeys,
                          size_t n, uint32_t *codes, int nthreads)
{
    ref_vjob j = {hashes, sigs, pubkeys, codes, (ref_verify_fn)fn, 32};
    /* bign128Verify builds its process-global curve object on first use (bign128.c:34-88):
       take that hit once, single-threaded, before fanning out */
    if (n) codes[0] = j.f(hashes, sigs, pubkeys);
    orc_parallel_for(n, nthreads, ref_verify_range, &j);
}
