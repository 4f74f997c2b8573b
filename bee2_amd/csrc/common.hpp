// common.hpp -- shared host-side plumbing of libbee2hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "../../include/bee2hip.h"
#ifdef BEE2HIP_EXPERIMENTS      // test / A/B hooks: only in libbee2hip_exp.so (bee2_amd/csrc/Makefile, target exp)
#include "../../include/bee2hip_internal.h"
#endif

namespace bee2hip {

// record a HIP failure for bee2hip_last_error(); returns ERR_BEE2HIP_DEVICE
err_t hip_fail(hipError_t e, const char *what);

#define B2H_TRY(call)                                                   \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return ::bee2hip::hip_fail(e_, #call);    \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Library-owned device scratch, cached per (device, stream): two batches in flight on
// different streams never share a buffer, two on the same stream are ordered by the stream.
// Grown on demand; growing frees the old block only after the stream has drained.
err_t scratch_for_stream(hipStream_t st, int slot, size_t bytes, void **out);
// a library-owned second queue of the calling thread on the current device and the events of a fork / join around it (capi.hip)
err_t side_stream(hipStream_t *side, hipEvent_t *fork, hipEvent_t *join);
const uint8_t *host_beltH();                              // the belt S-box, generated once on the host (capi.hip)

// per-device launch facts and the once-per-(device, kernel) dynamic-LDS grant (belt_kernels.hip)
int cur_dev();
int num_cus();
hipError_t dyn_lds_once(const void *kern, size_t bytes);
err_t upload_beltH(const uint8_t *H);            // the S-box copy of bee2hip_tu_belt.hip
err_t upload_beltH_bign(const uint8_t *H);       // ... of bee2hip_tu_bign.hip

// ---- kernel launchers (defined next to their kernels) ----
err_t launch_bashF_batch(void *d_states, size_t n, hipStream_t st);
err_t launch_bashHash_beltMAC(const void *d_msgs, size_t msg_len, size_t n, size_t l,
                              const uint32_t key[8], bool do_hash, bool do_mac,
                              void *d_digests, void *d_tags, hipStream_t st);
// one lane absorbs `count` bytes into one sponge state (drop-in bashHashStepH)
err_t launch_belt_ctr_blocks(void *d_buf, size_t nblocks, const uint32_t key[8],
                             const uint32_t ctr0[4], uint64_t first, void *d_last_gamma,
                             hipStream_t st);
err_t launch_belt_encr_blocks(void *d_blocks, size_t nblocks, const uint32_t key[8], hipStream_t st);
err_t launch_belt_decr_blocks(void *d_blocks, size_t nblocks, const uint32_t key[8], hipStream_t st);
// mode: 0 ECB encrypt, 1 ECB decrypt (src may equal dst), 2 CBC decrypt (src != dst)
err_t launch_belt_modes(int mode, const void *d_src, void *d_dst, size_t nblocks, const uint32_t key[8],
                        const uint32_t iv[4], hipStream_t st);
err_t launch_belt_cbc_encr(void *d_msgs, size_t nblk, size_t n, const uint32_t key[8], void *d_ivs, hipStream_t st);
err_t launch_bign_verify(size_t l, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                         const void *d_sigs, const void *d_pubkeys, size_t n, void *d_codes,
                         hipStream_t st);

err_t launch_bash_sponge(void *d_states, const void *d_data, size_t stride, size_t count, size_t n,
                         int fin, hipStream_t st);
err_t launch_belt_mac(void *d_states, const void *d_data, size_t stride, size_t count, size_t n,
                      int mode, hipStream_t st);
// alg: 0 = belt-hash; 128 / 192 / 256 = bash256 / bash384 / bash512
err_t launch_belt_bde(int decr, const void *d_src, void *d_dst, size_t nblocks, const uint32_t key[8],
                      const uint32_t s[4], uint64_t first, void *d_s_out, hipStream_t st);
err_t launch_belt_sde(int decr, void *d_sectors, size_t nblk, size_t nsectors, const uint32_t key[8],
                      const void *d_ivs, hipStream_t st);
err_t launch_belt_che(const void *d_src, void *d_dst, size_t nblocks, const uint32_t key[8], const uint32_t s[4],
                      uint64_t first, void *d_s_out, hipStream_t st);
err_t launch_belt_polyhash(const void *d_data, size_t nbytes, const uint32_t r[4], const uint32_t t[4],
                           void *d_t_out, hipStream_t st);
err_t launch_bash_sponge_cols(void *d_state, const void *d_data, size_t nblocks, hipStream_t st);
err_t launch_belt_hash_stream(void *d_hs, const void *d_data, size_t nblocks, int fin, uint64_t bits_lo,
                              uint64_t bits_hi, hipStream_t st);
err_t launch_hash_ragged(size_t alg, const void *d_data, const void *d_off, const void *d_order, size_t n,
                         void *d_digests, hipStream_t st, bool secret = false);
err_t launch_bign_pubkey_val(size_t l, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st);
// n signatures under ONE key of a standard curve (bign_kernels.hip "one signer"): pubkey = HOST memory.  Returns
// ERR_KEY_NOT_ON_CURVE (not a bee2 code; never leaves the library) when the key is not a point of the curve: the caller
// then takes the general path, which like bee2 does not look (bign_sign.c:306-311).
constexpr err_t ERR_KEY_NOT_ON_CURVE = 0xFFFFFFF1u;
err_t launch_bign_verify_onekey(size_t l, const uint8_t *oid_der, size_t oid_len, const void *d_hashes, const void *d_sigs,
                                const uint8_t *pubkey, size_t n, void *d_codes, hipStream_t st);
// ... of nkeys signers: pubkeys HOST (nkeys keys), d_key_index n x uint32 on the device (an index >= nkeys: ERR_BAD_INPUT for that
// signature); a key off the curve costs its signatures the slow kernel, not the call
err_t launch_bign_verify_keyed(size_t l, const uint8_t *oid_der, size_t oid_len, const void *d_hashes, const void *d_sigs,
                               const uint8_t *pubkeys, size_t nkeys, const void *d_key_index, size_t n, void *d_codes, hipStream_t st);
err_t launch_replicate_key(const void *d_key, size_t key_bytes, size_t n, void *d_out, hipStream_t st);
unsigned long long bign_onekey_table_builds();
// non-standard parameter sets (bign_generic_kernels.hip): params already through bignParamsCheck's tests
err_t launch_bign_verify_generic(const bign_params *params, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                                 const void *d_sigs, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st);
err_t bign_generic_check(const bign_params *params);
err_t launch_bign_pubkey_val_generic(const bign_params *params, const void *d_pubkeys, size_t n, void *d_codes, hipStream_t st);
// 8f-4 tail: Q = d G per private key (codes: ERR_OK / ERR_BAD_PRIVKEY); signing, mode 0 = bignSign2 (d_aux = t or
// null), mode 1 = one-time keys supplied (d_aux = k)
err_t launch_bign_pubkey_calc_generic(const bign_params *params, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys,
                                      void *d_codes, hipStream_t st);
err_t launch_bign_sign_generic(const bign_params *params, int mode, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                               const void *d_privkeys, const void *d_aux, size_t t_len, int t_shared, size_t n, void *d_sigs,
                               void *d_codes, hipStream_t st);
err_t launch_bign_pubkey_calc(size_t l, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes, hipStream_t st);
err_t launch_bign_sign(size_t l, int mode, const uint8_t *oid_der, size_t oid_len, const void *d_hashes,
                       const void *d_privkeys, const void *d_aux, size_t t_len, int t_shared, size_t n, void *d_sigs,
                       void *d_codes, hipStream_t st);
err_t launch_bign_debug_fe(size_t l, int op, const void *a, const void *b, void *out, size_t n, hipStream_t st);

}  // namespace bee2hip
