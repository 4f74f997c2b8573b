// TEST INFRASTRUCTURE: a C view of bee2_amd/csrc/host_small.hpp (the drop-in layer's host path for small single
// calls) so that tests/test_host_small.py can pin every function to the oracle and the golden vectors on CPU,
// without a GPU.  Built by the test itself: g++ -O2 -shared -fPIC.  Nothing here ships.
#include "../../bee2_amd/csrc/host_small.hpp"

using namespace bee2hip::hostp;
static BeltTables g_T;

extern "C" {
void hs_init(const uint8_t H[256]) { belt_tables(g_T, H); }
void hs_bashF(uint8_t s[192]) { bashF(s); }
void hs_sponge(uint8_t s[192], size_t buf_len, size_t *pos, const uint8_t *buf, size_t count) { sponge_absorb(s, buf_len, pos, buf, count); }
void hs_encr(uint32_t x[4], const uint32_t K[8]) { belt_encr(g_T, x, K); }
void hs_decr(uint32_t x[4], const uint32_t K[8]) { belt_decr(g_T, x, K); }
void hs_ctr(uint8_t *buf, size_t count, const uint32_t key[8], uint32_t ctr[4], uint8_t block[16], size_t *reserved)
{
    ctr_blocks(g_T, buf, count, key, ctr, block, reserved);
}
void hs_mac(const uint32_t key[8], uint32_t s[4], uint32_t r[4], uint32_t mac[4], uint8_t block[16], size_t *filled,
            const uint8_t *buf, size_t count, int mode)
{
    mac_step(g_T, key, s, r, mac, block, filled, buf, count, mode);
}
void hs_hash(uint32_t hs[12], const uint8_t *data, size_t nblocks, int fin, uint64_t lo, uint64_t hi) { hash_stream(g_T, hs, data, nblocks, fin, lo, hi); }
void hs_polyhash(uint32_t t[4], const uint32_t r[4], const uint8_t *data, size_t nbytes) { polyhash(t, r, data, nbytes); }
// form 0 = what the product runs (PCLMULQDQ where the CPU has it), 1 = table of 16 multiples, 2 = bit-serial definition
void hs_polyhash_form(uint32_t t[4], const uint32_t r[4], const uint8_t *data, size_t nbytes, int form) { polyhash(t, r, data, nbytes, form); }
int hs_have_clmul(void) { return gf_have_clmul() ? 1 : 0; }
void hs_modes(int mode, uint8_t *buf, size_t nblocks, const uint32_t key[8], const uint32_t iv[4]) { modes_blocks(g_T, mode, buf, nblocks, key, iv); }
void hs_cbc_encr(uint8_t *buf, size_t nblocks, const uint32_t key[8], uint8_t chain[16]) { cbc_encr_blocks(g_T, buf, nblocks, key, chain); }
void hs_bde(int decr, uint8_t *buf, size_t nblocks, const uint32_t key[8], uint32_t s[4]) { bde_blocks(g_T, decr, buf, nblocks, key, s); }
void hs_che(uint8_t *buf, size_t nblocks, const uint32_t key[8], uint32_t s[4]) { che_blocks(g_T, buf, nblocks, key, s); }
void hs_sde(int decr, uint8_t *buf, size_t count, const uint8_t iv[16], const uint32_t key[8]) { sde_sector(g_T, decr, buf, count, iv, key); }
}
