// TEST INFRASTRUCTURE: a C view of bee2_amd/csrc/host_bign.hpp (the drop-in layer's host path for ONE signature
// verification) so that tests/test_host_bign.py can pin it to the reference's verdicts (committed fixtures) and to the
// oracle on CPU, without a GPU.  Built by the test itself: g++ -O2 -shared -fPIC.  Nothing here ships.
#include "../../bee2_amd/csrc/host_bign.hpp"
#include "../../bee2_amd/csrc/bign_curves.inc"

using namespace bee2hip;
static hostp::BeltTables g_T;
static uint8_t g_H[256];
static hostb::Curve<4> g_c128;
static hostb::Curve<6> g_c192;
static hostb::Curve<8> g_c256;

extern "C" {
void hb_init(const uint8_t H[256])
{
    memcpy(g_H, H, 256);
    hostp::belt_tables(g_T, H);
    g_c128.init(BIGN128_CRANDALL_C, k_bign128_q, k_bign128_yG);
    g_c192.init(BIGN192_CRANDALL_C, k_bign192_q, k_bign192_yG);
    g_c256.init(BIGN256_CRANDALL_C, k_bign256_q, k_bign256_yG);
}
uint32_t hb_verify(size_t l, const uint8_t *oid, size_t oid_len, const uint8_t *hash, const uint8_t *sig, const uint8_t *pubkey,
                   uint8_t *rx)
{
    if (l == 128) return hostb::verify<4>(g_c128, g_T, g_H, oid, oid_len, hash, sig, pubkey, rx);
    if (l == 192) return hostb::verify<6>(g_c192, g_T, g_H, oid, oid_len, hash, sig, pubkey, rx);
    return hostb::verify<8>(g_c256, g_T, g_H, oid, oid_len, hash, sig, pubkey, rx);
}
// field operations on 8 l / 64 little-endian words (op: 0 mul, 1 sqr, 2 add, 3 sub, 4 inv)
void hb_field(size_t l, int op, uint64_t *r, const uint64_t *a, const uint64_t *b)
{
#define HB_DO(N, C)                                                                        \
    {                                                                                      \
        hostb::Field<N> F{C};                                                              \
        hostb::Fe<N> x, y, z;                                                              \
        memcpy(x.v, a, 8 * N); memcpy(y.v, b, 8 * N);                                      \
        if (op == 0) F.mul(z, x, y); else if (op == 1) F.sqr(z, x); else if (op == 2) F.add(z, x, y);   \
        else if (op == 3) F.sub(z, x, y); else F.inv(z, x);                                \
        memcpy(r, z.v, 8 * N);                                                             \
    }
    if (l == 128) HB_DO(4, BIGN128_CRANDALL_C) else if (l == 192) HB_DO(6, BIGN192_CRANDALL_C) else HB_DO(8, BIGN256_CRANDALL_C)
#undef HB_DO
}
uint32_t hb_pubkey_val(size_t l, const uint8_t *pubkey)
{
    if (l == 128) return hostb::pubkey_val<4>(g_c128, k_bign128_b, pubkey);
    if (l == 192) return hostb::pubkey_val<6>(g_c192, k_bign192_b, pubkey);
    return hostb::pubkey_val<8>(g_c256, k_bign256_b, pubkey);
}
int hb_wnaf(int8_t *out, const uint64_t *k, int nl, int w) { return hostb::wnaf(out, k, nl, w); }
}
