/*
 * oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the three bee2 hot paths this repo accelerates
 * (bashF / bash hash, belt block / CTR / MAC / hash, bign verify on
 * bign-curve256v1).  It exists so that the GPU box -- which has no copy of the
 * reference -- still has an independent checker and a CPU baseline.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (bee2_amd/csrc, libbee2hip.so) never links it,
 * never calls it and has no CPU fallback.
 *
 * Pinning: every function below is checked bit-for-bit against the reference
 * itself (oracle/_ref/libbee2ref.so, built from /root/reference by
 * oracle/Makefile) and against the STB known-answer vectors committed under
 * tests/golden/ -- see tests/test_oracle_*.py.  Parity is therefore PINNED.
 *
 * All byte strings are little-endian octet strings exactly as in bee2.
 */
#ifndef BEE2_AMD_ORACLE_H
#define BEE2_AMD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bee2 err_t values used on the path (include/bee2/core/err.h:72-74,132,180,186,196) */
#define ORC_OK            0u
#define ORC_BAD_INPUT     109u
#define ORC_BAD_OID       301u
#define ORC_BAD_PARAMS    502u
#define ORC_BAD_PUBKEY    505u
#define ORC_BAD_MAC       511u   /* err.h:198 */
#define ORC_BAD_SIG       510u

/* ---- bash (STB 34.101.77) ------------------------------------------------ */
/* bashF: include/bee2/crypto/bash.h:136, src/crypto/bash/bash_f64.c:174-187 */
void orc_bashF(uint8_t block[192]);
/* n independent states, contiguous; nthreads >= 1 host threads */
void orc_bashF_batch(uint8_t *states, size_t n, int nthreads);
/* bashHash: src/crypto/bash/bash_hash.c:118-137.  l = security level (bits),
   digest = l/4 bytes.  Returns ORC_BAD_PARAMS for a bad level. */
uint32_t orc_bashHash(uint8_t *hash, size_t l, const uint8_t *src, size_t count);
/* incremental form (Start / StepH / StepG), state is caller-owned */
typedef struct {
    uint8_t s[192];
    size_t buf_len;
    size_t pos;
} orc_bash_hash_st;
void orc_bashHashStart(orc_bash_hash_st *st, size_t l);
void orc_bashHashStepH(const uint8_t *buf, size_t count, orc_bash_hash_st *st);
void orc_bashHashStepG(uint8_t *hash, size_t hash_len, const orc_bash_hash_st *st);

/* ---- belt (STB 34.101.31) ------------------------------------------------ */
const uint8_t *orc_beltH(void);                       /* belt_block.c:43-66 */
void orc_beltKeyExpand2(uint32_t key_[8], const uint8_t *key, size_t len); /* :88-106 */
void orc_beltBlockEncr2(uint32_t block[4], const uint32_t key[8]);         /* :323-327 */
void orc_beltBlockEncr(uint8_t block[16], const uint32_t key[8]);          /* :302-321 */

typedef struct {            /* belt_lcl.h:135-141 */
    uint32_t key[8];
    uint32_t ctr[4];
    uint8_t block[16];
    size_t reserved;
} orc_belt_ctr_st;
void orc_beltCTRStart(orc_belt_ctr_st *st, const uint8_t *key, size_t len, const uint8_t iv[16]);
void orc_beltCTRStepE(void *buf, size_t count, orc_belt_ctr_st *st);
uint32_t orc_beltCTR(void *dest, const void *src, size_t count,
                     const uint8_t *key, size_t len, const uint8_t iv[16]);
/* bulk, full blocks only, starting at block index `first` (1-based counter
   offset = first + i + 1), multi-threaded: the CPU baseline for H2 */
void orc_beltCTR_blocks(uint8_t *buf, size_t nblocks, const uint32_t key[8],
                        const uint32_t ctr0[4], uint64_t first, int nthreads);

typedef struct {            /* belt_mac.c:32-40 */
    uint32_t key[8];
    uint32_t s[4];
    uint32_t r[4];
    uint8_t block[16];
    size_t filled;
} orc_belt_mac_st;
void orc_beltMACStart(orc_belt_mac_st *st, const uint8_t *key, size_t len);
void orc_beltMACStepA(const void *buf, size_t count, orc_belt_mac_st *st);
void orc_beltMACStepG(uint8_t mac[8], const orc_belt_mac_st *st);
uint32_t orc_beltMAC(uint8_t mac[8], const void *src, size_t count,
                     const uint8_t *key, size_t len);

/* block decryption and the block-parallel modes built on it */
void orc_beltBlockDecr2(uint32_t block[4], const uint32_t key[8]);          /* belt_block.c:341-373 */
uint32_t orc_beltECB(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     int decr);                                              /* belt_ecb.c:52-159 */
uint32_t orc_beltCBC(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     const uint8_t iv[16], int decr);                        /* belt_cbc.c:63-193 */
uint32_t orc_beltBDE(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     const uint8_t iv[16], int decr);                        /* belt_bde.c:40-133 */
void orc_beltBDE_blocks(void *buf, size_t nblocks, const uint32_t key[8], uint32_t s[4], int decr); /* belt_bde.c:51-85 */
void orc_beltCHE_blocks(void *buf, size_t nblocks, const uint32_t key[8], uint32_t s[4]);           /* belt_che.c:86-98 */
uint32_t orc_beltWBL(void *buf, size_t nblocks, const uint32_t key[8], int decr); /* belt_wbl.c:58-152, whole blocks */
uint32_t orc_beltSDE(void *dest, const void *src, size_t count, const uint8_t *key, size_t len,
                     const uint8_t iv[16], int decr);                        /* belt_sde.c:38-121 */

/* belt-dwp, authenticated encryption (belt_dwp.c:27-274) */
typedef struct {
    orc_belt_ctr_st ctr;
    uint32_t r[4], t[4];
    uint64_t bits_open, bits_crit;
    uint8_t block[16];
    size_t filled;
} orc_belt_dwp_st;
void orc_beltDWPStart(orc_belt_dwp_st *st, const uint8_t *key, size_t len, const uint8_t iv[16]);
void orc_beltDWPStepE(void *buf, size_t count, orc_belt_dwp_st *st);            /* = StepD */
void orc_beltDWPStepI(const void *buf, size_t count, orc_belt_dwp_st *st);
void orc_beltDWPStepA(const void *buf, size_t count, orc_belt_dwp_st *st);
void orc_beltDWPStepG(uint8_t mac[8], const orc_belt_dwp_st *st);
uint32_t orc_beltDWPWrap(void *dest, uint8_t mac[8], const void *src1, size_t count1, const void *src2,
                         size_t count2, const uint8_t *key, size_t len, const uint8_t iv[16]);
uint32_t orc_beltDWPUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                           const uint8_t mac[8], const uint8_t *key, size_t len, const uint8_t iv[16]);

/* belt-che (belt_che.c:27-319): mac = the belt-dwp authenticator state (its ctr.key holds K; r = E_K(iv)) */
typedef struct {
    orc_belt_dwp_st mac;
    uint32_t s[4];
    uint8_t gamma[16];
    size_t reserved;
} orc_belt_che_st;
void orc_beltCHEStart(orc_belt_che_st *st, const uint8_t *key, size_t len, const uint8_t iv[16]);
void orc_beltCHEStepE(void *buf, size_t count, orc_belt_che_st *st);            /* = StepD */
void orc_beltCHEStepI(const void *buf, size_t count, orc_belt_che_st *st);
void orc_beltCHEStepA(const void *buf, size_t count, orc_belt_che_st *st);
void orc_beltCHEStepG(uint8_t mac[8], const orc_belt_che_st *st);
uint32_t orc_beltCHEWrap(void *dest, uint8_t mac[8], const void *src1, size_t count1, const void *src2,
                         size_t count2, const uint8_t *key, size_t len, const uint8_t iv[16]);
uint32_t orc_beltCHEUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                           const uint8_t mac[8], const uint8_t *key, size_t len, const uint8_t iv[16]);

void orc_beltCompr(uint32_t h[8], const uint32_t X[8]);        /* belt_compr.c:27-51 */
uint32_t orc_beltHash(uint8_t hash[32], const void *src, size_t count); /* belt_hash.c:173-190 */

/* ---- bign (STB 34.101.45), bign-curve256v1 only -------------------------- */
/* bign128Verify: src/crypto/bign/bign128.c:177-185 -> bignVerifyEc
   src/crypto/bign/bign_sign.c:268-347.  rx (optional, 32 bytes) receives LE(x_R)
   when the double-scalar multiplication produced a finite point. */
uint32_t orc_bign128Verify(const uint8_t hash[32], const uint8_t sig[48],
                           const uint8_t pubkey[64]);
uint32_t orc_bign128Verify_ex(const uint8_t hash[32], const uint8_t sig[48],
                              const uint8_t pubkey[64], uint8_t rx[32]);
void orc_bign128Verify_batch(const uint8_t *hashes, const uint8_t *sigs,
                             const uint8_t *pubkeys, size_t n, uint32_t *codes,
                             int nthreads);
/* any of the three standard curves.  l in {128,192,256}; hash l/4, sig 3l/8,
   pubkey l/2 octets.  bignVerify: src/crypto/bign/bign_sign.c:349-361 */
uint32_t orc_bignVerify_ex(size_t l, const uint8_t *oid_der, size_t oid_len, const uint8_t *hash,
                           const uint8_t *sig, const uint8_t *pubkey, uint8_t *rx);
/* bignPubkeyVal (bign_misc.c:319-365): pubkey l/2 octets -> ORC_OK / ORC_BAD_PUBKEY */
uint32_t orc_bignPubkeyVal(size_t l, const uint8_t *pubkey);
void orc_bignPubkeyVal_batch(size_t l, const uint8_t *pubkeys, size_t n, uint32_t *codes);
uint32_t orc_bign192Verify(const uint8_t hash[48], const uint8_t sig[72], const uint8_t pubkey[96]);
uint32_t orc_bign256Verify(const uint8_t hash[64], const uint8_t sig[96], const uint8_t pubkey[128]);
void orc_bignVerify_batch(size_t l, const uint8_t *oid_der, size_t oid_len, const uint8_t *hashes,
                          const uint8_t *sigs, const uint8_t *pubkeys, size_t n, uint32_t *codes,
                          int nthreads);

/* SURVEY.md 8f-4, second half: public key from private key, key generation, signing
   (bign_misc.c:182-229,373-417, bign_sign.c:32-245).  rng-driven entries take the rng's OUTPUT
   (consecutive l/4-octet draws) instead of a callback. */
uint32_t orc_bignPubkeyCalc(size_t l, uint8_t *pubkey, const uint8_t *privkey);
uint32_t orc_bignKeypairGen(size_t l, uint8_t *privkey, uint8_t *pubkey, const uint8_t *rnd, size_t ndraws, size_t *used);
uint32_t orc_bignSign_rnd(size_t l, uint8_t *sig, const uint8_t *oid_der, size_t oid_len, const uint8_t *hash,
                          const uint8_t *privkey, const uint8_t *rnd, size_t ndraws, size_t *used);
uint32_t orc_bignSign2(size_t l, uint8_t *sig, const uint8_t *oid_der, size_t oid_len, const uint8_t *hash,
                       const uint8_t *privkey, const void *t, size_t t_len);

/* ---- mixed bash512 + beltMAC per message (H4) ---------------------------- */
void orc_bash512_beltMAC_batch(const uint8_t *msgs, size_t msg_len, size_t n,
                               const uint8_t *key, size_t key_len,
                               uint8_t *digests /* n*64 */, uint8_t *tags /* n*8 */,
                               int nthreads);

/* ---- drivers that time the REFERENCE (oracle/_ref/libbee2ref*.so) on host threads:
   the caller passes the reference's own function pointers (bashF, beltCTRStepE,
   bign128Verify, bashHash, beltMAC) obtained with dlsym/ctypes ---------------------- */
void orc_drive_ref_bashF(void *bashF_fn, uint8_t *states, size_t n, int nthreads);
void orc_drive_ref_ctr(void *beltCTRStepE_fn, uint8_t *buf, size_t nblocks, const uint32_t key[8],
                       const uint32_t ctr0[4], uint64_t first, int nthreads);
void orc_drive_ref_verify(void *bign128Verify_fn, const uint8_t *hashes, const uint8_t *sigs,
                          const uint8_t *pubkeys, size_t n, uint32_t *codes, int nthreads);
void orc_drive_ref_mode(void *fn, const uint8_t *src, uint8_t *dst, size_t nblocks, const uint8_t *key,
                        size_t klen, const uint8_t *iv, int nthreads);
void orc_drive_ref_mixed(void *bashHash_fn, void *beltMAC_fn, const uint8_t *msgs, size_t msg_len,
                         size_t n, const uint8_t *key, size_t key_len, uint8_t *digests,
                         uint8_t *tags, int nthreads);

/* deterministic synthetic-input generator shared by tests and bench:
   x_i = splitmix64(seed + i), written little-endian */
void orc_fill_splitmix64(uint8_t *buf, size_t nbytes, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
