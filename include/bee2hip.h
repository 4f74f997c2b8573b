/*
 * bee2hip.h -- C ABI of libbee2hip.so, the MI355X (gfx950) batch-primitive engine
 * for bee2's data-parallel hot paths: bash-f / bashHash, belt (block, CTR, MAC, hash,
 * ECB, CBC, BDE, SDE, DWP, CHE) and bign signature verification on the three standard
 * curves.
 *
 * Three groups of entry points, all `extern "C"`, plain pointers and sizes:
 *
 *  (1) bee2 DROP-IN symbols: same names, signatures and error behaviour as the
 *      bee2 functions they replace, so a bee2 caller can link this library instead
 *      of those objects (bee2's own test/crypto sources pass against it:
 *      oracle/Makefile `reftests`).  States are caller-owned PODs of X_keep() bytes that
 *      may be memcpy-cloned, as in bee2; belt_ctr_st, belt_mac_st and bash_hash_st have
 *      bee2's exact layout, the other states are opaque.  Where a call runs: every large
 *      call and every bign operation that touches a private or one-time key evaluates its
 *      primitives (bashF, E_K / D_K, GF(2^128) products, the scalar multiplications) on the
 *      GPU; a SMALL single call -- one bashF, one block, a block-parallel mode under 8 KiB
 *      per call, the serial chain of one message (sponge, CBC-MAC, belt-hash, CBC
 *      encryption, one belt-sde sector), ONE signature verification on a standard curve --
 *      takes the library's own host path (bee2_amd/csrc/host_small.hpp, host_bign.hpp;
 *      SURVEY.md 8b "single-call path; may run on CPU"), because one launch costs 20 us
 *      where a host core needs 0.3 (one verification: 0.42 ms through the GPU, 27 us on
 *      the calling core, 178 us in bee2).  BEE2HIP_FORCE=gpu|cpu overrides the choice.  In
 *      every mode the calling thread needs a working HIP device: there is no GPU-less
 *      operation.  A device failure inside a `void` function is retried once and then
 *      finished on the host with a message on stderr (BEE2HIP_FORCE=gpu: abort with a
 *      message).  Each declaration cites the bee2 interface it replaces.  All of them
 *      may be called from several threads at once.
 *
 *  (2) bee2hip_*  host-pointer batch API: the caller hands host buffers; the
 *      library stages H2D, launches, stages D2H.
 *
 *  (3) bee2hip_*_dev  device-pointer batch API: buffers already resident in HBM
 *      (e.g. torch tensors' data_ptr()); asynchronous on `stream` (a hipStream_t
 *      passed as void*, NULL = default stream).  This is what bench.py times.
 *
 * All octet strings are little-endian, exactly as in bee2.
 */
#ifndef BEE2HIP_H
#define BEE2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)      /* everything declared here is exported */
#endif

/* ---- bee2 base types (include/bee2/defs.h:269,441,463,520) ---------------- */
typedef uint8_t octet;
typedef uint32_t u32;
typedef int bool_t;
typedef uint32_t err_t;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif

/* ---- bee2 error codes on the path (include/bee2/core/err.h) --------------- */
#define ERR_OK               ((err_t)0)
#define ERR_BAD_INPUT        ((err_t)109)   /* err.h:72  */
#define ERR_OUTOFMEMORY      ((err_t)110)   /* err.h:74  */
#define ERR_NOT_IMPLEMENTED  ((err_t)119)   /* err.h:92  */
#define ERR_FILE_NOT_FOUND    ((err_t)202)   /* err.h:105 */
#define ERR_BAD_OID          ((err_t)301)   /* err.h:132 */
#define ERR_BAD_RNG          ((err_t)304)   /* err.h:138 */
#define ERR_BAD_PARAMS       ((err_t)502)   /* err.h:180 */
#define ERR_BAD_PRIVKEY      ((err_t)504)   /* err.h:184 */
#define ERR_BAD_PUBKEY       ((err_t)505)   /* err.h:186 */
#define ERR_BAD_SIG          ((err_t)510)   /* err.h:196 */
#define ERR_BAD_MAC          ((err_t)511)   /* err.h:198 */
/* engine-specific: a HIP runtime call failed (no bee2 equivalent; the reference
   has no device).  bee2hip_last_error() returns the HIP message. */
#define ERR_BEE2HIP_DEVICE   ((err_t)0x4850)

/* ======================================================================== *
 * (1) bee2 drop-in symbols
 * ======================================================================== */

/* ---- bash: include/bee2/crypto/bash.h ------------------------------------ */
/* bash.h:136  (the BASH_PLATFORM slot, src/crypto/bash/bash_f.c:26-44) */
void bashF(octet block[192], void *stack);
/* bash.h:139 */
size_t bashF_deep(void);
/* src/crypto/bash/bash_f.c:27-43, consumed by test/crypto/bash_bench.c:27,47 */
extern const char bash_platform[];
/* bash.h:152-225, src/crypto/bash/bash_hash.c:25-137 */
size_t bashHash_keep(void);
void bashHashStart(void *state, size_t l);
void bashHashStepH(const void *buf, size_t count, void *state);
void bashHashStepG(octet hash[], size_t hash_len, void *state);
bool_t bashHashStepV(const octet hash[], size_t hash_len, void *state);
err_t bashHash(octet hash[], size_t l, const void *src, size_t count);

/* ---- belt: include/bee2/crypto/belt.h ------------------------------------ */
/* belt.h:148, src/crypto/belt/belt_block.c:62-66 */
const octet *beltH(void);
/* belt.h:170-176, belt_block.c:88-106 */
void beltKeyExpand2(u32 key_[8], const octet key[], size_t len);
/* belt.h:186-221, belt_block.c:302-334 */
void beltBlockEncr(octet block[16], const u32 key[8]);
void beltBlockEncr2(u32 block[4], const u32 key[8]);
void beltBlockEncr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8]);
/* belt.h:692-743, src/crypto/belt/belt_ctr.c:46-135; state = belt_ctr_st,
   src/crypto/belt/belt_lcl.h:135-141 (key[8], ctr[4], block[16], reserved) */
size_t beltCTR_keep(void);
void beltCTRStart(void *state, const octet key[], size_t len, const octet iv[16]);
void beltCTRStepE(void *buf, size_t count, void *state);
#define beltCTRStepD beltCTRStepE                 /* belt.h:724 */
err_t beltCTR(void *dest, const void *src, size_t count, const octet key[], size_t len,
              const octet iv[16]);
/* ---- SURVEY.md 8f-1 ("next" row): block decryption and the block-parallel modes ---- */
/* belt.h:230-262, belt_block.c:341-373 */
void beltBlockDecr(octet block[16], const u32 key[8]);
void beltBlockDecr2(u32 block[4], const u32 key[8]);
void beltBlockDecr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8]);
/* belt.h:430-505, src/crypto/belt/belt_ecb.c:42-159; state = belt_ecb_st (key[8], block[16]) */
size_t beltECB_keep(void);
void beltECBStart(void *state, const octet key[], size_t len);
void beltECBStepE(void *buf, size_t count, void *state);
void beltECBStepD(void *buf, size_t count, void *state);
err_t beltECBEncr(void *dest, const void *src, size_t count, const octet key[], size_t len);
err_t beltECBDecr(void *dest, const void *src, size_t count, const octet key[], size_t len);
/* belt.h:520-600, src/crypto/belt/belt_cbc.c:43-193; state = belt_cbc_st (key[8], block[16], block1[16]) */
size_t beltCBC_keep(void);
void beltCBCStart(void *state, const octet key[], size_t len, const octet iv[16]);
void beltCBCStepE(void *buf, size_t count, void *state);
void beltCBCStepD(void *buf, size_t count, void *state);
err_t beltCBCEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                  const octet iv[16]);
err_t beltCBCDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                  const octet iv[16]);

/* belt-bde, block-wise disk encryption (belt.h:1360-1455, src/crypto/belt/belt_bde.c:26-133):
   s <- E_K(iv); for every 16-byte block s <- s*x in GF(2^128), Y = E_K(X ^ s) ^ s (StepD: D_K).
   count must be a multiple of 16; the one-shots return ERR_BAD_INPUT for count < 16 or
   count % 16 != 0 (belt_bde.c:93-100). */
size_t beltBDE_keep(void);
void beltBDEStart(void *state, const octet key[], size_t len, const octet iv[16]);
void beltBDEStepE(void *buf, size_t count, void *state);
void beltBDEStepD(void *buf, size_t count, void *state);
err_t beltBDEEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                  const octet iv[16]);
err_t beltBDEDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                  const octet iv[16]);

/* belt-sde, sector-wise disk encryption (belt.h, src/crypto/belt/belt_sde.c:26-121): XEX around
   the wide-block cipher belt-wbl with the tweak E_K(iv) on the first block.  One call = one sector:
   count a multiple of 16, >= 32 (the one-shots return ERR_BAD_INPUT otherwise, belt_sde.c:79-80).
   A sector is a serial chain of 2*(count/16) block encryptions; the batch entry below runs many
   sectors, one lane each. */
size_t beltSDE_keep(void);
void beltSDEStart(void *state, const octet key[], size_t len);
void beltSDEStepE(void *buf, size_t count, const octet iv[16], void *state);
void beltSDEStepD(void *buf, size_t count, const octet iv[16], void *state);
err_t beltSDEEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                  const octet iv[16]);
err_t beltSDEDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                  const octet iv[16]);

/* belt-dwp, authenticated encryption: CTR + polynomial MAC over GF(2^128) (belt.h, src/crypto/belt/
   belt_dwp.c:27-274).  The state is an opaque POD of beltDWP_keep() bytes (not bee2's layout: bee2
   appends a beltPolyMul stack).  Order of calls as in bee2: Start, StepI* (open data), then
   StepE + StepA (protect) or StepA + StepD (unprotect) on the critical data, StepG / StepV at any
   point.  beltDWPUnwrap returns ERR_BAD_MAC without decrypting when the tag does not match. */
size_t beltDWP_keep(void);
void beltDWPStart(void *state, const octet key[], size_t len, const octet iv[16]);
void beltDWPStepE(void *buf, size_t count, void *state);
void beltDWPStepI(const void *buf, size_t count, void *state);
void beltDWPStepA(const void *buf, size_t count, void *state);
void beltDWPStepD(void *buf, size_t count, void *state);
void beltDWPStepG(octet mac[8], void *state);
bool_t beltDWPStepV(const octet mac[8], void *state);
err_t beltDWPWrap(void *dest, octet mac[8], const void *src1, size_t count1, const void *src2,
                  size_t count2, const octet key[], size_t len, const octet iv[16]);
err_t beltDWPUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                    const octet mac[8], const octet key[], size_t len, const octet iv[16]);

/* belt-che (belt_che.c:27-319): the belt-dwp authenticator with r = E_K(iv) and the keystream
   E_K(s_i), s_0 = r, s_i = s_{i-1}*x ^ 1 in GF(2^128); same call order and error behaviour as belt-dwp */
size_t beltCHE_keep(void);
void beltCHEStart(void *state, const octet key[], size_t len, const octet iv[16]);
void beltCHEStepE(void *buf, size_t count, void *state);
void beltCHEStepI(const void *buf, size_t count, void *state);
void beltCHEStepA(const void *buf, size_t count, void *state);
void beltCHEStepD(void *buf, size_t count, void *state);
void beltCHEStepG(octet mac[8], void *state);
bool_t beltCHEStepV(const octet mac[8], void *state);
err_t beltCHEWrap(void *dest, octet mac[8], const void *src1, size_t count1, const void *src2,
                  size_t count2, const octet key[], size_t len, const octet iv[16]);
err_t beltCHEUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                    const octet mac[8], const octet key[], size_t len, const octet iv[16]);

/* belt-hash, the default algorithm of `bee2cmd bsum` (belt.h, src/crypto/belt/belt_hash.c:28-190;
   cmd/bsum/bsum.c:133-147 drives exactly Start / StepH* / StepG).  One message is a serial chain:
   these calls are for drop-in use, batches go through bee2hip_hash_ragged. */
size_t beltHash_keep(void);
void beltHashStart(void *state);
void beltHashStepH(const void *buf, size_t count, void *state);
void beltHashStepG(octet hash[32], void *state);
void beltHashStepG2(octet hash[], size_t hash_len, void *state);
bool_t beltHashStepV(const octet hash[32], void *state);
bool_t beltHashStepV2(const octet hash[], size_t hash_len, void *state);
err_t beltHash(octet hash[32], const void *src, size_t count);

/* belt.h:756-854, src/crypto/belt/belt_mac.c:32-203 */
size_t beltMAC_keep(void);
void beltMACStart(void *state, const octet key[], size_t len);
void beltMACStepA(const void *buf, size_t count, void *state);
void beltMACStepG(octet mac[8], void *state);
void beltMACStepG2(octet mac[], size_t mac_len, void *state);
bool_t beltMACStepV(const octet mac[8], void *state);
bool_t beltMACStepV2(const octet mac[], size_t mac_len, void *state);
err_t beltMAC(octet mac[8], const void *src, size_t count, const octet key[], size_t len);

/* ---- bign: include/bee2/crypto/bign.h, bign128.h -------------------------- */
/* bign.h:65-74 */
typedef struct {
    size_t l;
    octet p[64];
    octet a[64];
    octet b[64];
    octet q[64];
    octet yG[64];
    octet seed[8];
} bign_params;
/* bign.h:100-107, src/crypto/bign/bign_params.c:180-230: "1.2.112.0.2.0.34.101.45.3.{1,2,3}" */
err_t bignParamsStd(bign_params *params, const char *name);
/* bign.h:395-402, src/crypto/bign/bign_sign.c:349-361.
   Any parameter set that passes bignParamsCheck + bignEcCreate is served (bignVerify and bignPubkeyVal; batch forms
   likewise): the three standard sets by the throughput kernels, every other set by general-curve kernels (Montgomery
   arithmetic, general coefficient a; 14 / 35 / 72 ms per batch of up to a few thousand signatures on the three
   levels -- a completeness path, not a throughput path).  The signing side (bignKeypairGen, bignPubkeyCalc, bignSign*)
   serves them too: a constant-time double-and-add-always ladder with complete additions on the general curve, arithmetic
   mod q in Montgomery form (bign_generic_kernels.hip; tests/golden/bign_generic_sign.json holds the reference's answers).
   Any valid DER OID is served, as in bee2: beyond 128 octets its leading whole 32-byte blocks are belt-hashed once per
   batch and the per-signature kernels continue from that state (tests/golden/bign_oid_long.json: 129 .. 4099 octets). */
err_t bignVerify(const bign_params *params, const octet oid_der[], size_t oid_len,
                 const octet hash[], const octet sig[], const octet pubkey[]);
/* include/bee2/crypto/bign128.h:174-178, src/crypto/bign/bign128.c:177-185 */
err_t bign128Verify(const octet hash[32], const octet sig[48], const octet pubkey[64]);
/* the 384- and 512-bit curves (l = 192 / 256).
   include/bee2/crypto/bign192.h, bign256.h; src/crypto/bign/bign192.c:177-185, bign256.c:177-185 */
err_t bign192Verify(const octet hash[48], const octet sig[72], const octet pubkey[96]);
err_t bign256Verify(const octet hash[64], const octet sig[96], const octet pubkey[128]);
/* public-key validation, the step `bee2cmd sig vfy` runs before each verification
   (cmd/core/cmd_sig.c:463-478): bign.h:278-281, src/crypto/bign/bign_misc.c:319-365 (coordinates < p and
   the point on the curve, src/math/ecp/ecp_a.c:36-60); bign128.h:87, bign192.h, bign256.h facades
   (bign128.c:115-122).  ERR_OK or ERR_BAD_PUBKEY. */
err_t bignPubkeyVal(const bign_params *params, const octet pubkey[]);
err_t bign128PubkeyVal(const octet pubkey[64]);
err_t bign192PubkeyVal(const octet pubkey[96]);
err_t bign256PubkeyVal(const octet pubkey[128]);

/* SURVEY.md 8f-4, second half: key generation, public key from private key, signing.  Secret-handling device code:
   constant-time at the instruction level (bee2_amd/csrc/bign_sign_kernels.hip states what that covers).
   gen_i: include/bee2/defs.h (void (*)(void* buf, size_t count, void* state)), called on the HOST exactly as
   zzRandNZMod calls it (src/math/zz/zz_mod.c:463-485); the scalar multiplications, the nonce derivation of
   bignSign2 (belt-hash + belt-wbl), the hash tail and the arithmetic mod q run on the device.
   bign.h:207-215,247-254,316-326,352-363; bign128.h:64-69,98-101,140-164 (and bign192.h, bign256.h);
   src/crypto/bign/bign_misc.c:182-243,373-431, bign_sign.c:32-260. */
typedef void (*gen_i)(void *buf, size_t count, void *state);
err_t bignKeypairGen(octet privkey[], octet pubkey[], const bign_params *params, gen_i rng, void *rng_state);
err_t bignPubkeyCalc(octet pubkey[], const bign_params *params, const octet privkey[]);
err_t bignSign(octet sig[], const bign_params *params, const octet oid_der[], size_t oid_len, const octet hash[],
               const octet privkey[], gen_i rng, void *rng_state);
err_t bignSign2(octet sig[], const bign_params *params, const octet oid_der[], size_t oid_len, const octet hash[],
                const octet privkey[], const void *t, size_t t_len);
err_t bign128KeypairGen(octet privkey[32], octet pubkey[64], gen_i rng, void *rng_state);
err_t bign128PubkeyCalc(octet pubkey[64], const octet privkey[32]);
err_t bign128Sign(octet sig[48], const octet hash[32], const octet privkey[32], gen_i rng, void *rng_state);
err_t bign128Sign2(octet sig[48], const octet hash[32], const octet privkey[32], const void *t, size_t t_len);
err_t bign192KeypairGen(octet privkey[48], octet pubkey[96], gen_i rng, void *rng_state);
err_t bign192PubkeyCalc(octet pubkey[96], const octet privkey[48]);
err_t bign192Sign(octet sig[72], const octet hash[48], const octet privkey[48], gen_i rng, void *rng_state);
err_t bign192Sign2(octet sig[72], const octet hash[48], const octet privkey[48], const void *t, size_t t_len);
err_t bign256KeypairGen(octet privkey[64], octet pubkey[128], gen_i rng, void *rng_state);
err_t bign256PubkeyCalc(octet pubkey[128], const octet privkey[64]);
err_t bign256Sign(octet sig[96], const octet hash[64], const octet privkey[64], gen_i rng, void *rng_state);
err_t bign256Sign2(octet sig[96], const octet hash[64], const octet privkey[64], const void *t, size_t t_len);

/* ======================================================================== *
 * (2) host-pointer batch API (new; SURVEY.md 8b "batch extension")
 * ======================================================================== */

/* n independent 192-byte states, contiguous, permuted in place (batched bashF) */
err_t bee2hip_bashF_batch(octet *states, size_t n);
/* bulk CTR over host memory, exactly equivalent to beltCTRStepE(buf, count, state)
   including the ctr / block / reserved fields it leaves behind (belt_ctr.c:66-111) */
err_t bee2hip_beltCTR_bulk(void *buf, size_t count, void *ctr_state);
/* n signatures: hashes n*32, sigs n*48, pubkeys n*64 -> codes[n] = what
   bignVerify(params, oid_der, oid_len, ...) returns per item (bign_sign.c:268-361) */
err_t bee2hip_bignVerify_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                               const octet *hashes, const octet *sigs, const octet *pubkeys,
                               size_t n, err_t *codes);
/* n public keys of params->l/2 octets each -> codes[n] = what bignPubkeyVal(params, key) returns */
err_t bee2hip_bignPubkeyVal_batch(const bign_params *params, const octet *pubkeys, size_t n,
                                  err_t *codes);
/* n private keys of params->l/4 octets -> n public keys of params->l/2 octets; codes[n] = what bignPubkeyCalc
   returns per key (ERR_OK / ERR_BAD_PRIVKEY); the public key of a refused private key is left untouched */
err_t bee2hip_bignPubkeyCalc_batch(const bign_params *params, const octet *privkeys, size_t n,
                                   octet *pubkeys, err_t *codes);
/* n deterministic signatures (bignSign2): hashes n*(l/4), privkeys n*(l/4), one additional input t (may be
   NULL) shared by the batch -> sigs n*(3l/8), codes[n] = what bignSign2 returns per item */
err_t bee2hip_bignSign2_batch(const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                              const octet *privkeys, const void *t, size_t t_len, size_t n, octet *sigs, err_t *codes);
/* n signatures with the one-time keys supplied (what bignSign computes once its rng has produced k in
   {1..q-1}): ks n*(l/4); a k outside that range gives ERR_BAD_RNG for the item */
err_t bee2hip_bignSignK_batch(const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                              const octet *privkeys, const octet *ks, size_t n, octet *sigs, err_t *codes);
/* n messages of msg_len bytes each (contiguous): bashHash(l) digest (l/4 bytes each,
   digests may be NULL) and beltMAC tag (8 bytes each, tags may be NULL) per message */
err_t bee2hip_bashHash_beltMAC_batch(const octet *msgs, size_t msg_len, size_t n, size_t l,
                                     const octet key[], size_t key_len,
                                     octet *digests, octet *tags);

/* SURVEY.md 8f-3 (batch front-end for `bee2cmd bsum`, cmd/bsum/bsum.c:133-221): n messages of
   different lengths packed back to back, message i = data[offsets[i] .. offsets[i+1]).
   alg = 0: belt-hash (32-byte digests, src/crypto/belt/belt_hash.c:173-190);
   alg = 128 / 192 / 256: bash256 / bash384 / bash512 (alg/4-byte digests, bash_hash.c:118-137) */
err_t bee2hip_hash_ragged(size_t alg, const octet *data, const uint64_t *offsets, size_t n,
                          octet *digests);
err_t bee2hip_hash_ragged_dev(size_t alg, const void *d_data, const void *d_offsets, size_t n,
                              void *d_digests, void *stream);
/* (without an order the library buckets the messages by the power of two of their length on the device
   and launches the buckets longest first)
   same with an explicit launch order: d_order = n x uint32, a permutation of 0..n-1 (NULL = as above);
   slot k hashes message d_order[k], digest i still lands at d_digests + i*dlen.  A slot is one lane,
   or for messages of 4 KiB and more a group of 8 lanes (bash: one column of the state each) or a
   pair (belt-hash: the two independent encryptions of a compression).  A wavefront runs until the
   longest of its messages is done: pass the messages sorted by decreasing length
   (bee2hip_hash_ragged does so itself). */
err_t bee2hip_hash_ragged_ordered_dev(size_t alg, const void *d_data, const void *d_offsets,
                                      const void *d_order, size_t n, void *d_digests, void *stream);

/* ---- one host batch over several GPUs from one process --------------------------------
   The batch is cut into contiguous index ranges (bee2hip_multi_plan), one worker thread per device runs the
   single-device entry above on its range; no data crosses devices.  ndev = number of devices to use, 0 = all
   visible ones (bee2hip_device_count).  Results are those of the single-device call, item for item; the CTR
   state is left exactly as bee2hip_beltCTR_bulk leaves it. */
int bee2hip_device_count(void);
err_t bee2hip_multi_plan(size_t n, int parts, int i, size_t *first, size_t *count);
err_t bee2hip_bashF_batch_multi(octet *states, size_t n, int ndev);
err_t bee2hip_beltCTR_bulk_multi(void *buf, size_t count, void *ctr_state, int ndev);
err_t bee2hip_bignVerify_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                     const octet *hashes, const octet *sigs, const octet *pubkeys, size_t n,
                                     err_t *codes, int ndev);
err_t bee2hip_bignVerify_onekey_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                            const octet *hashes, const octet *sigs, const octet pubkey[], size_t n,
                                            err_t *codes, int ndev);
err_t bee2hip_bignVerify_keyed_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                           const octet *hashes, const octet *sigs, const octet *pubkeys, size_t nkeys,
                                           const u32 *key_index, size_t n, err_t *codes, int ndev);
err_t bee2hip_bignSign2_batch_multi(const bign_params *params, const octet oid_der[], size_t oid_len,
                                    const octet *hashes, const octet *privkeys, const void *t, size_t t_len, size_t n,
                                    octet *sigs, err_t *codes, int ndev);
err_t bee2hip_bashHash_beltMAC_batch_multi(const octet *msgs, size_t msg_len, size_t n, size_t l, const octet key[],
                                           size_t key_len, octet *digests, octet *tags, int ndev);
err_t bee2hip_hash_ragged_multi(size_t alg, const octet *data, const uint64_t *offsets, size_t n, octet *digests,
                                int ndev);

/* ======================================================================== *
 * (3) device-pointer batch API (buffers in HBM; async on `stream`)
 *     Device pointers to states, blocks, sectors, messages and verify records must be 16-byte
 *     aligned (the kernels read them as 16-byte vectors), d_codes 4-byte, ragged offsets 8-byte and digests 4-byte:
 *     a misaligned pointer is refused with ERR_BAD_INPUT.  One stream = one queue: threads that
 *     share a stream serialise their calls themselves.
 * ======================================================================== */
err_t bee2hip_bashF_batch_dev(void *d_states, size_t n, void *stream);
/* full 16-byte blocks only: block i (0-based) ^= E_K(ctr0 + first_block + i + 1),
   the 128-bit little-endian counter arithmetic of belt_ctr.c:27-35 */
err_t bee2hip_beltCTR_blocks_dev(void *d_buf, size_t nblocks, const u32 key[8],
                                 const u32 ctr0[4], uint64_t first_block, void *stream);
/* ECB-style E_K over n blocks in place (used for ctr0 = E_K(iv), r = E_K(0)) */
err_t bee2hip_beltBlockEncr_dev(void *d_blocks, size_t nblocks, const u32 key[8], void *stream);
/* 8f-1: full blocks, device resident.  mode 0 = ECB encrypt, 1 = ECB decrypt (d_src may equal
   d_dst), 2 = CBC decrypt with chaining value iv[4] (u32 words; d_src != d_dst) */
err_t bee2hip_beltModes_blocks_dev(int mode, const void *d_src, void *d_dst, size_t nblocks,
                                   const u32 key[8], const u32 iv[4], void *stream);
/* n independent messages of nblk full blocks each, CBC-encrypted in place, one lane per
   message; d_ivs[n][16] holds each message's iv on entry and its last ciphertext block on exit */
err_t bee2hip_beltCBCEncr_batch_dev(void *d_msgs, size_t nblk, size_t n, const u32 key[8],
                                    void *d_ivs, void *stream);
/* belt-bde on nblocks whole blocks, device resident, d_src may equal d_dst.  s[4] = E_K(iv) as
   u32 words (what beltBDEStart leaves in the state); the blocks are the ones first_block ..
   first_block + nblocks - 1 of the stream, i.e. block j uses the tweak s * x^(j+1) -- a stream can
   be cut into pieces (or sharded across GPUs) at any block boundary.  decr = 0 / 1.
   d_s_out (may be NULL) receives s * x^(first_block + nblocks), 16 bytes: the state after the piece. */
/* belt-dwp's authenticator on device-resident data: *d_t_out (16 bytes, device) <- the value of t
   after absorbing nbytes at d_data as 16-byte blocks, t <- (t ^ X) * r in GF(2^128), the last block
   zero-padded (belt_dwp.c:96-101,118-119); r[4], t[4] = u32 words as in the state.  nbytes = 0
   returns t.  Together with bee2hip_beltCTR_blocks_dev this is beltDWPWrap on resident data. */
err_t bee2hip_beltDWP_absorb_dev(const void *d_data, size_t nbytes, const u32 r[4], const u32 t[4],
                                 void *d_t_out, void *stream);
/* belt-sde on nsectors contiguous sectors of sector_bytes each (a multiple of 16, >= 32), device
   resident, in place; d_ivs = nsectors x 16 bytes, one iv per sector.  decr = 0 / 1. */
err_t bee2hip_beltSDE_sectors_dev(int decr, void *d_sectors, size_t sector_bytes, size_t nsectors,
                                  const u32 key[8], const void *d_ivs, void *stream);
/* belt-che keystream on nblocks whole blocks, device resident, d_src may equal d_dst: block j of the
   stream (first_block <= j < first_block + nblocks) is XORed with E_K(S_{j+1}), S_0 = s[4] = E_K(iv) as
   u32 words, S_i = S_{i-1}*x ^ 1.  d_s_out (may be NULL) receives S_{first_block + nblocks}, 16 bytes. */
err_t bee2hip_beltCHE_blocks_dev(const void *d_src, void *d_dst, size_t nblocks, const u32 key[8],
                                 const u32 s[4], uint64_t first_block, void *d_s_out, void *stream);
err_t bee2hip_beltBDE_blocks_dev(int decr, const void *d_src, void *d_dst, size_t nblocks,
                                 const u32 key[8], const u32 s[4], uint64_t first_block,
                                 void *d_s_out, void *stream);
err_t bee2hip_bign128Verify_batch_dev(const void *d_hashes, const void *d_sigs,
                                      const void *d_pubkeys, size_t n, void *d_codes,
                                      void *stream);
err_t bee2hip_bignVerify_batch_dev(const octet oid_der[], size_t oid_len,
                                   const void *d_hashes, const void *d_sigs,
                                   const void *d_pubkeys, size_t n, void *d_codes,
                                   void *stream);
/* same for security level l in {128, 192, 256}: hashes n*(l/4), sigs n*(3l/8), pubkeys n*(l/2) */
err_t bee2hip_bignVerifyL_batch_dev(size_t l, const octet oid_der[], size_t oid_len,
                                    const void *d_hashes, const void *d_sigs,
                                    const void *d_pubkeys, size_t n, void *d_codes, void *stream);
/* n signatures under ONE public key -- the batch `bee2cmd sig vfy` makes of a tree of files signed by one party
   (cmd/core/cmd_sig.c:484-490), a signed log, a package repository.  Same verdict per signature as
   bignVerify(params, oid, hash_i, sig_i, pubkey) (bign_sign.c:268-361), in about a fifth of the work: the key becomes a
   fixed base with a comb table of its own (built once per key and device, cached: the last 1024 keys), so the scalar multiplication has no
   doublings left.  pubkey is HOST memory in both forms (l/2 octets).  A key that is not a point of the curve, and a
   non-standard parameter set, take the general path with the key repeated -- same codes, general speed.
   The _dev form synchronises `stream` the first time it meets a key (table construction); later calls with that key
   only queue kernels (hipGraph capture: prime the key with one call before capturing). */
err_t bee2hip_bignVerify_onekey_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                      const octet *hashes, const octet *sigs, const octet pubkey[],
                                      size_t n, err_t *codes);
err_t bee2hip_bignVerifyL_onekey_batch_dev(size_t l, const octet oid_der[], size_t oid_len,
                                           const void *d_hashes, const void *d_sigs, const octet pubkey[],
                                           size_t n, void *d_codes, void *stream);
/* ... and of a FEW signers: pubkeys = nkeys keys (HOST memory, nkeys <= 4096 on the device form), key_index[i] < nkeys says whose
   signature i is (n x u32; device memory in the _dev form).  codes[i] = bignVerify(params, oid, hash_i, sig_i, pubkeys[key_index[i]]);
   an index out of range gives ERR_BAD_INPUT for that signature.  Every key gets its cached 8-bit comb table (the last 1024 keys
   per process) and, once it has been busy enough, the 16-bit one; a key that is not a point of the curve costs ITS signatures the
   complete slow kernel, nothing else.  This form uploads keys and table addresses per call (a copy and a synchronisation). */
err_t bee2hip_bignVerify_keyed_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                     const octet *hashes, const octet *sigs, const octet *pubkeys, size_t nkeys,
                                     const u32 *key_index, size_t n, err_t *codes);
err_t bee2hip_bignVerifyL_keyed_batch_dev(size_t l, const octet oid_der[], size_t oid_len,
                                          const void *d_hashes, const void *d_sigs, const octet pubkeys[], size_t nkeys,
                                          const void *d_key_index, size_t n, void *d_codes, void *stream);
/* n public keys of l/2 octets, l in {128, 192, 256}; d_pubkeys 16-byte aligned, d_codes n x err_t */
/* 8f-4 tail, device resident (private keys, one-time keys and t 4-byte aligned, hashes 16-byte).  d_codes as the
   host forms; outputs of refused items are zero.  t: n x t_len octets, or one string of t_len octets when
   t_shared != 0 (t_len <= 64 on this path; NULL = none) */
err_t bee2hip_bignPubkeyCalcL_batch_dev(size_t l, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes,
                                        void *stream);
err_t bee2hip_bignSign2L_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                   const void *d_privkeys, const void *d_t, size_t t_len, int t_shared, size_t n,
                                   void *d_sigs, void *d_codes, void *stream);
err_t bee2hip_bignSignKL_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                   const void *d_privkeys, const void *d_ks, size_t n, void *d_sigs, void *d_codes,
                                   void *stream);
err_t bee2hip_bignPubkeyValL_batch_dev(size_t l, const void *d_pubkeys, size_t n, void *d_codes,
                                       void *stream);
err_t bee2hip_bashHash_beltMAC_batch_dev(const void *d_msgs, size_t msg_len, size_t n, size_t l,
                                         const octet key[], size_t key_len,
                                         void *d_digests, void *d_tags, void *stream);

/* Shards that already LIVE on the devices: entry i of each array belongs to device i (a pointer into its memory and an item
   count); worker i launches the single-device _dev entry on its own non-blocking stream of that device, ordered behind
   everything the device's NULL stream had been given when the call was made (a shard produced on the default stream or on a
   blocking stream needs no synchronisation; one produced on a NON-blocking stream must be complete), and the call returns when
   all are done.  Shards that share a card therefore run side by side.  Nothing is staged through the host and nothing crosses devices.  ndev = number of array entries,
   1 <= ndev <= bee2hip_device_count().  CTR: shard i holds blocks first_block + nblocks[0] + .. + nblocks[i-1] onwards of
   ONE stream with counter ctr0 (as bee2hip_beltCTR_blocks_dev); MAC / hash: d_digests or d_tags may be NULL as a whole. */
err_t bee2hip_bashF_batch_multi_dev(void *const d_states[], const size_t counts[], int ndev);
err_t bee2hip_beltCTR_blocks_multi_dev(void *const d_bufs[], const size_t nblocks[], const u32 key[8],
                                       const u32 ctr0[4], uint64_t first_block, int ndev);
err_t bee2hip_bignVerifyL_batch_multi_dev(size_t l, const octet oid_der[], size_t oid_len, const void *const d_hashes[],
                                          const void *const d_sigs[], const void *const d_pubkeys[],
                                          const size_t counts[], void *const d_codes[], int ndev);
/* one signer / a few signers, shards resident on the GPUs (the keys on the host, as in the single-device forms) */
err_t bee2hip_bignVerifyL_onekey_batch_multi_dev(size_t l, const octet oid_der[], size_t oid_len, const void *const d_hashes[],
                                                 const void *const d_sigs[], const octet pubkey[], const size_t counts[],
                                                 void *const d_codes[], int ndev);
err_t bee2hip_bignVerifyL_keyed_batch_multi_dev(size_t l, const octet oid_der[], size_t oid_len, const void *const d_hashes[],
                                                const void *const d_sigs[], const octet pubkeys[], size_t nkeys,
                                                const void *const d_key_index[], const size_t counts[], void *const d_codes[],
                                                int ndev);
err_t bee2hip_bashHash_beltMAC_batch_multi_dev(const void *const d_msgs[], size_t msg_len, const size_t counts[], size_t l,
                                               const octet key[], size_t key_len, void *const d_digests[],
                                               void *const d_tags[], int ndev);

/* ---- engine management --------------------------------------------------- */
/* bind the calling thread's engine to HIP device `device` (default: current) */
err_t bee2hip_set_device(int device);
/* block until everything queued on `stream` is done */
err_t bee2hip_sync(void *stream);
/* text of the last HIP failure on this thread ("" if none) */
const char *bee2hip_last_error(void);
/* "bee2hip <ver> gfx950" */
const char *bee2hip_version(void);
/* path of the drop-in layer's SMALL single calls (one permutation, one block, one message's serial chain, one
   signature verification): 0 = by size (default), 1 = every primitive in a kernel, 2 = the host path wherever one
   exists.  The run-time form of the environment variable BEE2HIP_FORCE=gpu|cpu; mode < 0 only queries.  Returns the
   mode that was in force.  The _dev and _multi entry points and the batch entries are never affected: they always launch
   kernels -- with ONE exception: the host-pointer bee2hip_hash_ragged hands its few longest messages (>= 64 KiB, one serial
   chain each) to host threads while the GPU takes the rest, unless the mode is 1; those messages are not uploaded, and the
   call counts once in bee2hip_path_count(0). */
int bee2hip_path_policy(int mode);
/* how many drop-in helper calls of this process went where: which = 0 host path, 1 GPU path, 2 finished on the host
   after the GPU path failed twice (a warning is printed for each of those) */
unsigned long long bee2hip_path_count(int which);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BEE2HIP_H */
