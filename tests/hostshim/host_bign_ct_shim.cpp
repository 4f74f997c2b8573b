// TEST INFRASTRUCTURE: a C view of bee2_amd/csrc/host_bign_ct.hpp (the drop-in layer's constant-time host path for ONE
// key generation / public-key calculation / signature) so that tests/test_host_bign_ct.py can pin it on CPU to the
// reference's fixtures and to the oracle, and tools/ct_audit_x86.py can disassemble it.  Built by the test itself:
// g++ -O2 -shared -fPIC.  Nothing here ships.
#include "../../bee2_amd/csrc/host_bign_ct.hpp"
#include "../../bee2_amd/csrc/bign_curves.inc"

using namespace bee2hip;
static hostp::BeltTables g_T;
static uint8_t g_H[256];
static hostb::Curve<4> g_c128;
static hostb::Curve<6> g_c192;
static hostb::Curve<8> g_c256;
static hostct::SignCurve<4> g_s128;
static hostct::SignCurve<6> g_s192;
static hostct::SignCurve<8> g_s256;

extern "C" {
int hc_init(const uint8_t H[256])
{
    memcpy(g_H, H, 256);
    hostp::belt_tables(g_T, H);
    g_c128.init(BIGN128_CRANDALL_C, k_bign128_q, k_bign128_yG);
    g_c192.init(BIGN192_CRANDALL_C, k_bign192_q, k_bign192_yG);
    g_c256.init(BIGN256_CRANDALL_C, k_bign256_q, k_bign256_yG);
    g_s128.init(g_c128);
    g_s192.init(g_c192);
    g_s256.init(g_c256);
    return g_s128.ready && g_s192.ready && g_s256.ready;
}
uint32_t hc_pubkey_calc(size_t l, int keygen, const uint8_t *priv, uint8_t *pub)
{
    if (l == 128) return hostct::pubkey_calc<4>(g_s128, keygen != 0, priv, pub);
    if (l == 192) return hostct::pubkey_calc<6>(g_s192, keygen != 0, priv, pub);
    return hostct::pubkey_calc<8>(g_s256, keygen != 0, priv, pub);
}
uint32_t hc_sign(size_t l, const uint8_t *oid, size_t oid_len, const uint8_t *hash, const uint8_t *priv, const uint8_t *k,
                 const uint8_t *t, size_t t_len, uint8_t *sig)
{
    if (l == 128) return hostct::sign<4>(g_s128, g_T, g_H, oid, oid_len, hash, priv, k, t, t_len, sig);
    if (l == 192) return hostct::sign<6>(g_s192, g_T, g_H, oid, oid_len, hash, priv, k, t, t_len, sig);
    return hostct::sign<8>(g_s256, g_T, g_H, oid, oid_len, hash, priv, k, t, t_len, sig);
}
// field operations of the constant-time flavour on 8 l / 64 words (op: 0 mul, 2 add, 3 sub, 4 inv, 5 canon); the result is
// canonicalised unless op == 6 (raw add) / 7 (raw sub): the weakly reduced value itself
void hc_field(size_t l, int op, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
#define HC_DO(N, C)                                                                        \
    {                                                                                      \
        hostct::FieldCt<N> F{C};                                                           \
        hostb::Fe<N> x, y, z;                                                              \
        memcpy(x.v, a, 8 * N); memcpy(y.v, b, 8 * N);                                      \
        if (op == 0) F.mul(z, x, y); else if (op == 2 || op == 6) F.add(z, x, y);          \
        else if (op == 3 || op == 7) F.sub(z, x, y); else if (op == 4) F.inv(z, x); else z = x;      \
        if (op < 6) F.canon(z, z);                                                         \
        memcpy(r, z.v, 8 * N);                                                             \
    }
    if (l == 128) HC_DO(4, BIGN128_CRANDALL_C) else if (l == 192) HC_DO(6, BIGN192_CRANDALL_C) else HC_DO(8, BIGN256_CRANDALL_C)
#undef HC_DO
}
// cycles (rdtsc) of ONE k G with its range check and output, for tools/ct_audit_x86.py's timing comparison of key classes
uint64_t hc_time_pubkey_calc(size_t l, const uint8_t *priv, uint8_t *pub)
{
    unsigned lo, hi, lo2, hi2;
    __asm__ volatile("lfence\n rdtsc" : "=a"(lo), "=d"(hi) :: "memory");
    (void)hc_pubkey_calc(l, 1, priv, pub);
    __asm__ volatile("lfence\n rdtsc" : "=a"(lo2), "=d"(hi2) :: "memory");
    return (((uint64_t)hi2 << 32) | lo2) - (((uint64_t)hi << 32) | lo);
}
// x mod q for a 2N-limb x
void hc_mod_q(size_t l, uint64_t *r, const uint64_t *x)
{
    if (l == 128) { uint64_t rr[4], xx[8]; memcpy(xx, x, 64); g_s128.mod_q(rr, xx); memcpy(r, rr, 32); }
    else if (l == 192) { uint64_t rr[6], xx[12]; memcpy(xx, x, 96); g_s192.mod_q(rr, xx); memcpy(r, rr, 48); }
    else { uint64_t rr[8], xx[16]; memcpy(xx, x, 128); g_s256.mod_q(rr, xx); memcpy(r, rr, 64); }
}
}
