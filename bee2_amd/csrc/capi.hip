// capi.hip -- the C ABI of libbee2hip.so (declared in include/bee2hip.h).
//
// Host side of the engine: argument checks and state bookkeeping mirror bee2's
// C functions line for line in *behaviour* (same names, same state layouts, same
// error codes); every primitive evaluation is a kernel launch.  There is no CPU
// implementation of bashF / E_K / EC arithmetic in this library.
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.hpp"

namespace bee2hip {

// ------------------------------------------------------------------ errors ---
static thread_local char t_err[256] = "";

err_t hip_fail(hipError_t e, const char *what)
{
    snprintf(t_err, sizeof t_err, "%s: %s", what, hipGetErrorString(e));
    return ERR_BEE2HIP_DEVICE;
}

// ---------------------------------------------------------- per-device init ---
// the belt S-box from the standard's own generator (belt_block.c:21-35): an 8-bit
// LFSR stepped 116 times per entry, anchored at H[10] = 0x00, H[11] = 0x8E.
static uint8_t g_H[256];
static std::once_flag g_H_once;
static void gen_H()
{
    g_H[10] = 0x00; g_H[11] = 0x8E;
    for (unsigned x = 12; x < 10 + 256; ++x) {
        unsigned t = g_H[(x - 1) % 256];
        for (int i = 0; i < 116; ++i) t = (t >> 1) | ((unsigned)__builtin_parity(t & 0x63) << 7);
        g_H[x % 256] = (uint8_t)t;
    }
}
const uint8_t *host_beltH()
{
    std::call_once(g_H_once, gen_H);
    return g_H;
}

extern err_t upload_beltH(const uint8_t *H);       // belt_kernels.hip

constexpr int MAX_DEV = 64;
static std::mutex g_dev_mu;
static bool g_dev_ready[MAX_DEV];

// make sure the current device has its constants (S-box, curve tables)
err_t ensure_device()
{
    int dev = 0;
    B2H_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return ERR_BAD_INPUT;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_dev_ready[dev]) return ERR_OK;
    err_t code = upload_beltH(host_beltH());
    if (code != ERR_OK) return code;
    g_dev_ready[dev] = true;
    return ERR_OK;
}

// scratch device buffer for the host-pointer API, grown on demand, per thread
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int dev = -1;
    err_t need(size_t n)
    {
        int cur = 0;
        B2H_TRY(hipGetDevice(&cur));
        if (p && (cur != dev || cap < n)) { (void)hipFree(p); p = nullptr; cap = 0; }
        if (!p) {
            size_t want = n < 4096 ? 4096 : n;
            if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return ERR_OUTOFMEMORY; }
            cap = want; dev = cur;
        }
        return ERR_OK;
    }
    ~Scratch() { /* process teardown: the runtime may already be gone; leak deliberately */ }
};
static thread_local Scratch t_scr[4];

}  // namespace bee2hip

using namespace bee2hip;

// ============================================================== management ===
extern "C" err_t bee2hip_set_device(int device)
{
    B2H_TRY(hipSetDevice(device));
    return ensure_device();
}
extern "C" err_t bee2hip_sync(void *stream)
{
    B2H_TRY(hipStreamSynchronize(as_stream(stream)));
    return ERR_OK;
}
extern "C" const char *bee2hip_last_error(void) { return t_err; }
extern "C" const char *bee2hip_version(void) { return "bee2hip 0.1 gfx950"; }

// ===================================================== device-pointer batch ===
extern "C" err_t bee2hip_bashF_batch_dev(void *d_states, size_t n, void *stream)
{
    if (n && !d_states) return ERR_BAD_INPUT;
    return launch_bashF_batch(d_states, n, as_stream(stream));
}

extern "C" err_t bee2hip_beltCTR_blocks_dev(void *d_buf, size_t nblocks, const u32 key[8],
                                            const u32 ctr0[4], uint64_t first_block, void *stream)
{
    if ((nblocks && !d_buf) || !key || !ctr0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_ctr_blocks(d_buf, nblocks, key, ctr0, first_block, nullptr, as_stream(stream));
}

extern "C" err_t bee2hip_beltBlockEncr_dev(void *d_blocks, size_t nblocks, const u32 key[8], void *stream)
{
    if ((nblocks && !d_blocks) || !key) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_encr_blocks(d_blocks, nblocks, key, as_stream(stream));
}

// ======================================================= host-pointer batch ===
extern "C" err_t bee2hip_bashF_batch(octet *states, size_t n)
{
    if (n == 0) return ERR_OK;
    if (!states) return ERR_BAD_INPUT;
    Scratch &s = t_scr[0];
    err_t code = s.need(n * 192);
    if (code != ERR_OK) return code;
    B2H_TRY(hipMemcpy(s.p, states, n * 192, hipMemcpyHostToDevice));
    code = launch_bashF_batch(s.p, n, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(hipMemcpy(states, s.p, n * 192, hipMemcpyDeviceToHost));
    return ERR_OK;
}

// E_K over host blocks (n small): the only way the drop-in layer evaluates belt
static err_t encr_host_blocks(uint32_t *blocks, size_t n, const u32 key[8])
{
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    Scratch &s = t_scr[1];
    code = s.need(n * 16);
    if (code != ERR_OK) return code;
    B2H_TRY(hipMemcpy(s.p, blocks, n * 16, hipMemcpyHostToDevice));
    code = launch_belt_encr_blocks(s.p, n, key, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(hipMemcpy(blocks, s.p, n * 16, hipMemcpyDeviceToHost));
    return ERR_OK;
}

// a device failure inside a void bee2 function cannot be reported through the bee2
// signature: fail loudly instead of returning wrong bytes.
static void die_on(err_t code, const char *where)
{
    if (code == ERR_OK) return;
    fprintf(stderr, "libbee2hip: %s failed (err %u): %s\n", where, (unsigned)code, t_err);
    abort();
}

// =================================================================== bash ====
extern "C" const char bash_platform[] = "BASH_HIP_GFX950";

extern "C" void bashF(octet block[192], void *stack)
{
    (void)stack;                                   // bashF_deep() == 0
    die_on(bee2hip_bashF_batch(block, 1), "bashF");
}
extern "C" size_t bashF_deep(void) { return 0; }

// ==================================================================== belt ===
extern "C" const octet *beltH(void) { return host_beltH(); }

static inline u32 load32le(const octet *p)
{
    return (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
}
static inline void store32le(octet *p, u32 v)
{
    p[0] = (octet)v; p[1] = (octet)(v >> 8); p[2] = (octet)(v >> 16); p[3] = (octet)(v >> 24);
}

extern "C" void beltKeyExpand2(u32 key_[8], const octet key[], size_t len)
{
    // pure data formatting, no cipher work (belt_block.c:88-106)
    for (size_t i = 0; i < len / 4; ++i) key_[i] = load32le(key + 4 * i);
    if (len == 16) {
        key_[4] = key_[0]; key_[5] = key_[1]; key_[6] = key_[2]; key_[7] = key_[3];
    } else if (len == 24) {
        key_[6] = key_[0] ^ key_[1] ^ key_[2];
        key_[7] = key_[3] ^ key_[4] ^ key_[5];
    }
}

extern "C" void beltBlockEncr2(u32 block[4], const u32 key[8])
{
    die_on(encr_host_blocks(block, 1, key), "beltBlockEncr2");
}
extern "C" void beltBlockEncr(octet block[16], const u32 key[8])
{
    u32 w[4];
    for (int i = 0; i < 4; ++i) w[i] = load32le(block + 4 * i);
    beltBlockEncr2(w, key);
    for (int i = 0; i < 4; ++i) store32le(block + 4 * i, w[i]);
}
extern "C" void beltBlockEncr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8])
{
    u32 w[4] = {*a, *b, *c, *d};
    beltBlockEncr2(w, key);
    *a = w[0]; *b = w[1]; *c = w[2]; *d = w[3];
}

// ---- CTR: belt_ctr.c:46-135, state layout belt_lcl.h:135-141 ----
struct belt_ctr_st {
    u32 key[8];
    u32 ctr[4];
    octet block[16];
    size_t reserved;
};

extern "C" size_t beltCTR_keep(void) { return sizeof(belt_ctr_st); }

extern "C" void beltCTRStart(void *state, const octet key[], size_t len, const octet iv[16])
{
    belt_ctr_st *st = (belt_ctr_st *)state;
    beltKeyExpand2(st->key, key, len);
    for (int i = 0; i < 4; ++i) st->ctr[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->ctr, st->key);              // ctr0 = E_K(iv) on the GPU
    st->reserved = 0;
}

static inline void ctr_add(u32 c[4], uint64_t add)
{
    uint64_t lo = (uint64_t)c[0] | (uint64_t)c[1] << 32, hi = (uint64_t)c[2] | (uint64_t)c[3] << 32;
    const uint64_t nlo = lo + add;
    hi += nlo < lo;
    c[0] = (u32)nlo; c[1] = (u32)(nlo >> 32); c[2] = (u32)hi; c[3] = (u32)(hi >> 32);
}

extern "C" err_t bee2hip_beltCTR_bulk(void *buf_, size_t count, void *ctr_state)
{
    belt_ctr_st *st = (belt_ctr_st *)ctr_state;
    octet *buf = (octet *)buf_;
    if (!st || (count && !buf)) return ERR_BAD_INPUT;
    // gamma left over from the previous call (belt_ctr.c:70-83)
    if (st->reserved) {
        const size_t take = st->reserved < count ? st->reserved : count;
        const octet *g = st->block + 16 - st->reserved;
        for (size_t i = 0; i < take; ++i) buf[i] ^= g[i];
        st->reserved -= take; buf += take; count -= take;
        if (!count) return ERR_OK;
    }
    // whole blocks plus, if the tail is partial, one more gamma block: all on the GPU.
    // The tail is staged zero-padded to a full block; the kernel also hands back the
    // gamma of the final block, which the streaming state keeps (belt_ctr.c:89-96,101-108).
    const size_t full = count / 16, tail = count % 16;
    const size_t nblk = full + (tail ? 1 : 0);
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    Scratch &s = t_scr[2];
    code = s.need(nblk * 16 + 16);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (tail) B2H_TRY(hipMemset(d + full * 16, 0, 16));
    B2H_TRY(hipMemcpy(d, buf, count, hipMemcpyHostToDevice));
    // first_block = 0: the offset is relative to the state's *current* counter
    code = launch_belt_ctr_blocks(d, nblk, st->key, st->ctr, 0, d + nblk * 16, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(hipMemcpy(buf, d, count, hipMemcpyDeviceToHost));
    B2H_TRY(hipMemcpy(st->block, d + nblk * 16, 16, hipMemcpyDeviceToHost));
    ctr_add(st->ctr, nblk);                        // what nblk beltBlockIncU32 calls leave
    st->reserved = tail ? 16 - tail : 0;
    return ERR_OK;
}

extern "C" void beltCTRStepE(void *buf, size_t count, void *state)
{
    die_on(bee2hip_beltCTR_bulk(buf, count, state), "beltCTRStepE");
}

extern "C" err_t beltCTR(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16])
{
    if ((len != 16 && len != 24 && len != 32) || (count && (!src || !dest)) || !key || !iv)
        return ERR_BAD_INPUT;
    belt_ctr_st *st = new (std::nothrow) belt_ctr_st;
    if (!st) return ERR_OUTOFMEMORY;
    beltCTRStart(st, key, len, iv);
    memmove(dest, src, count);
    err_t code = bee2hip_beltCTR_bulk(dest, count, st);
    delete st;
    return code;
}
