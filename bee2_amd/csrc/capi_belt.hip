// capi_belt.hip -- belt: block, CTR, MAC, ECB / CBC / BDE / SDE, DWP / CHE, belt-hash; drop-ins (belt.h) and their batch entries.
// Part of the C ABI (capi.hip).
// ==================================================================== belt ===
extern "C" const octet *beltH(void) try { return host_beltH(); } catch (...) { (void)::bee2hip::caught(); return nullptr; }

static inline u32 load32le(const octet *p)
{
    return (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
}
static inline void store32le(octet *p, u32 v)
{
    p[0] = (octet)v; p[1] = (octet)(v >> 8); p[2] = (octet)(v >> 16); p[3] = (octet)(v >> 24);
}

extern "C" void beltKeyExpand2(u32 key_[8], const octet key[], size_t len)
try {
    // pure data formatting, no cipher work (belt_block.c:88-106)
    for (size_t i = 0; i < len / 4; ++i) key_[i] = load32le(key + 4 * i);
    if (len == 16) {
        key_[4] = key_[0]; key_[5] = key_[1]; key_[6] = key_[2]; key_[7] = key_[3];
    } else if (len == 24) {
        key_[6] = key_[0] ^ key_[1] ^ key_[2];
        key_[7] = key_[3] ^ key_[4] ^ key_[5];
    }
} B2H_CATCH_VOID("beltKeyExpand2")

extern "C" void beltBlockEncr2(u32 block[4], const u32 key[8])
try {
    die_on(encr_host_blocks(block, 1, key), "beltBlockEncr2");
} B2H_CATCH_VOID("beltBlockEncr2")
extern "C" void beltBlockEncr(octet block[16], const u32 key[8])
try {
    u32 w[4];
    for (int i = 0; i < 4; ++i) w[i] = load32le(block + 4 * i);
    beltBlockEncr2(w, key);
    for (int i = 0; i < 4; ++i) store32le(block + 4 * i, w[i]);
} B2H_CATCH_VOID("beltBlockEncr")
extern "C" void beltBlockEncr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8])
try {
    u32 w[4] = {*a, *b, *c, *d};
    beltBlockEncr2(w, key);
    *a = w[0]; *b = w[1]; *c = w[2]; *d = w[3];
} B2H_CATCH_VOID("beltBlockEncr3")

// ---- CTR: belt_ctr.c:46-135, state layout belt_lcl.h:135-141 ----
struct belt_ctr_st {
    u32 key[8];
    u32 ctr[4];
    octet block[16];
    size_t reserved;
};

extern "C" size_t beltCTR_keep(void) { return sizeof(belt_ctr_st); }

extern "C" void beltCTRStart(void *state, const octet key[], size_t len, const octet iv[16])
try {
    belt_ctr_st *st = (belt_ctr_st *)state;
    beltKeyExpand2(st->key, key, len);
    for (int i = 0; i < 4; ++i) st->ctr[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->ctr, st->key);              // ctr0 = E_K(iv) on the GPU
    st->reserved = 0;
} B2H_CATCH_VOID("beltCTRStart")

static inline void ctr_add(u32 c[4], uint64_t add)
{
    uint64_t lo = (uint64_t)c[0] | (uint64_t)c[1] << 32, hi = (uint64_t)c[2] | (uint64_t)c[3] << 32;
    const uint64_t nlo = lo + add;
    hi += nlo < lo;
    c[0] = (u32)nlo; c[1] = (u32)(nlo >> 32); c[2] = (u32)hi; c[3] = (u32)(hi >> 32);
}

// allow_host: the bee2 drop-ins (beltCTRStepE, beltCTR, beltDWPStepE ...) may finish a small call on the host; the batch
// entry point bee2hip_beltCTR_bulk never does
static err_t ctr_bulk(void *buf_, size_t count, void *ctr_state, bool allow_host)
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
    const auto gpu = [&]() -> err_t {
        if (count >= DUPLEX_MIN) {
            // all but the last (at most one) chunk through the duplex pipeline, whole blocks; what is left -- with the partial
            // block and the gamma the state keeps -- takes the plain path below, from the advanced counter
            const size_t CH = (size_t)1 << g_duplex_log2_blocks;             // blocks per chunk (2^20 = 16 MiB)
            const size_t pipe_blocks = (count - 1) / (16 * CH) * CH;
            err_t pc = ensure_device();
            if (pc != ERR_OK) return pc;
            Scratch &ps = t_scr[2];
            pc = ps.need(pipe_blocks * 16);
            if (pc != ERR_OK) return pc;
            const u32 *key = st->key, *ctr = st->ctr;
            size_t done = 0;
            pc = duplex_inplace(buf, (octet *)ps.p, 16, pipe_blocks, CH, [key, ctr](octet *d, size_t first, size_t cnt, hipStream_t s2) {
                return launch_belt_ctr_blocks(d, cnt, key, ctr, first, nullptr, s2);
            }, &done);
            // the blocks that came back ARE encrypted in the caller's buffer, also when a later chunk failed: whoever goes on
            // (the retry, the host fallback of with_host) starts behind them, from the advanced counter
            ctr_add(st->ctr, done);
            buf += done * 16;
            count -= done * 16;
            if (pc != ERR_OK) return pc;
        }
        const size_t full = count / 16, tail = count % 16;
        const size_t nblk = full + (tail ? 1 : 0);
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[2];
        code = s.need(nblk * 16 + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)s.p;
        if (tail) B2H_TRY(zero_staging(d + full * 16, 16));
        B2H_TRY(h2d(d, buf, count));
        // first_block = 0: the offset is relative to the state's *current* counter
        code = launch_belt_ctr_blocks(d, nblk, st->key, st->ctr, 0, d + nblk * 16, nullptr);
        if (code != ERR_OK) return code;
        octet last[16];
        B2H_TRY(d2h(last, d + nblk * 16, 16));
        B2H_TRY(d2h(buf, d, count));
        memcpy(st->block, last, 16);
        ctr_add(st->ctr, nblk);                        // what nblk beltBlockIncU32 calls leave
        st->reserved = tail ? 16 - tail : 0;
        return ERR_OK;
    };
    if (!allow_host) return gpu();
    return with_host(K_PARALLEL, count, "beltCTRStepE", gpu,
                     [&] { hostp::ctr_blocks(hostT(), buf, count, st->key, st->ctr, st->block, &st->reserved); });
}
extern "C" err_t bee2hip_beltCTR_bulk(void *buf, size_t count, void *ctr_state) try { return ctr_bulk(buf, count, ctr_state, false); } B2H_CATCH

extern "C" void beltCTRStepE(void *buf, size_t count, void *state)
try {
    die_on(ctr_bulk(buf, count, state, true), "beltCTRStepE");
} B2H_CATCH_VOID("beltCTRStepE")

extern "C" err_t beltCTR(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16])
try {
    if ((len != 16 && len != 24 && len != 32) || (count && (!src || !dest)) || !key || !iv)
        return ERR_BAD_INPUT;
    belt_ctr_st st[1];                                // (72 bytes: no blob, nothing to free on any path)
    beltCTRStart(st, key, len, iv);
    if (dest != src) memmove(dest, src, count);
    const err_t code = ctr_bulk(dest, count, st, true);
    wipe_host(st, sizeof st);
    return code;
} B2H_CATCH


// ================================================================ belt MAC ===
extern "C" size_t beltMAC_keep(void) { return sizeof(belt_mac_st); }

static err_t mac_host(belt_mac_st *st, const octet *buf, size_t count, int mode)
{
    return with_host(K_SERIAL, count, "beltMAC", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[1];
        code = s.need(sizeof(belt_mac_st) + 8 + count + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)s.p;
        const size_t off = (sizeof(belt_mac_st) + 15) & ~(size_t)15;
        B2H_TRY(h2d(d, st, sizeof *st));
        if (count) B2H_TRY(h2d(d + off, buf, count));
        code = launch_belt_mac(d, d + off, 0, count, 1, mode, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(st, d, sizeof *st));
        return ERR_OK;
    }, [&] { hostp::mac_step(hostT(), st->key, st->s, st->r, st->mac, st->block, &st->filled, buf, count, mode); });
}

extern "C" void beltMACStart(void *state, const octet key[], size_t len)
try {
    belt_mac_st *st = (belt_mac_st *)state;
    beltKeyExpand2(st->key, key, len);
    die_on(mac_host(st, nullptr, 0, 1), "beltMACStart");     // s = 0, r = E_K(0), filled = 0
} B2H_CATCH_VOID("beltMACStart")

extern "C" void beltMACStepA(const void *buf, size_t count, void *state)
try {
    belt_mac_st *st = (belt_mac_st *)state;
    // still filling the look-ahead block: no cipher work (belt_mac.c:63-70)
    if (st->filled < 16 && count <= 16 - st->filled) {
        memcpy(st->block + st->filled, buf, count);
        st->filled += count;
        return;
    }
    die_on(mac_host(st, (const octet *)buf, count, 2), "beltMACStepA");
} B2H_CATCH_VOID("beltMACStepA")

extern "C" void beltMACStepG2(octet mac[], size_t mac_len, void *state)
try {
    belt_mac_st *st = (belt_mac_st *)state;
    die_on(mac_host(st, nullptr, 0, 4), "beltMACStepG");
    octet full[8];
    store32le(full, st->mac[0]);
    store32le(full + 4, st->mac[1]);
    memcpy(mac, full, mac_len);
} B2H_CATCH_VOID("beltMACStepG2")
extern "C" void beltMACStepG(octet mac[8], void *state) try { beltMACStepG2(mac, 8, state); } B2H_CATCH_VOID("beltMACStepG")

extern "C" bool_t beltMACStepV2(const octet mac[], size_t mac_len, void *state)
try {
    octet full[8];
    beltMACStepG2(full, 8, state);
    return memcmp(mac, full, mac_len) == 0;
} B2H_CATCH_FALSE("beltMACStepV2")
extern "C" bool_t beltMACStepV(const octet mac[8], void *state) try { return beltMACStepV2(mac, 8, state); } B2H_CATCH_FALSE("beltMACStepV")

extern "C" err_t beltMAC(octet mac[8], const void *src, size_t count, const octet key[], size_t len)
try {
    if ((len != 16 && len != 24 && len != 32) || (count && !src) || !key || !mac) return ERR_BAD_INPUT;
    belt_mac_st *st = new (std::nothrow) belt_mac_st;
    if (!st) return ERR_OUTOFMEMORY;
    beltMACStart(st, key, len);
    beltMACStepA(src, count, st);
    beltMACStepG(mac, st);
    delete st;
    return ERR_OK;
} B2H_CATCH


// ============================================ 8f-1: block decrypt, ECB, CBC ===
static err_t decr_host_blocks(uint32_t *blocks, size_t n, const u32 key[8])
{
    return with_host(K_PRIM, n * 16, "belt block decryption", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[1];
        code = s.need(n * 16);
        if (code != ERR_OK) return code;
        B2H_TRY(h2d(s.p, blocks, n * 16));
        code = launch_belt_decr_blocks(s.p, n, key, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(blocks, s.p, n * 16));
        return ERR_OK;
    }, [&] { for (size_t i = 0; i < n; ++i) hostp::belt_decr(hostT(), blocks + 4 * i, key); });
}

extern "C" void beltBlockDecr2(u32 block[4], const u32 key[8])
try {
    die_on(decr_host_blocks(block, 1, key), "beltBlockDecr2");
} B2H_CATCH_VOID("beltBlockDecr2")
extern "C" void beltBlockDecr(octet block[16], const u32 key[8])
try {
    u32 w[4];
    for (int i = 0; i < 4; ++i) w[i] = load32le(block + 4 * i);
    beltBlockDecr2(w, key);
    for (int i = 0; i < 4; ++i) store32le(block + 4 * i, w[i]);
} B2H_CATCH_VOID("beltBlockDecr")
extern "C" void beltBlockDecr3(u32 *a, u32 *b, u32 *c, u32 *d, const u32 key[8])
try {
    u32 w[4] = {*a, *b, *c, *d};
    beltBlockDecr2(w, key);
    *a = w[0]; *b = w[1]; *c = w[2]; *d = w[3];
} B2H_CATCH_VOID("beltBlockDecr3")

extern "C" err_t bee2hip_beltModes_blocks_dev(int mode, const void *d_src, void *d_dst, size_t nblocks,
                                              const u32 key[8], const u32 iv[4], void *stream)
try {
    if (misaligned(d_src, 16) || misaligned(d_dst, 16)) return ERR_BAD_INPUT;
    if ((nblocks && (!d_src || !d_dst)) || !key || (mode == 2 && !iv)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_modes(mode, d_src, d_dst, nblocks, key, iv, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_beltCBCEncr_batch_dev(void *d_msgs, size_t nblk, size_t n, const u32 key[8],
                                               void *d_ivs, void *stream)
try {
    if (misaligned(d_msgs, 16) || misaligned(d_ivs, 16)) return ERR_BAD_INPUT;
    if ((n && (!d_msgs || !d_ivs)) || !key) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_cbc_encr(d_msgs, nblk, n, key, d_ivs, as_stream(stream));
} B2H_CATCH

// whole blocks of a host buffer through one of the block-parallel modes
static err_t modes_host(int mode, octet *buf, size_t nblocks, const u32 key[8], const octet chain[16])
{
    if (nblocks == 0) return ERR_OK;
    const size_t bytes = nblocks * 16;
    u32 iv[4] = {0, 0, 0, 0};
    if (chain) for (int i = 0; i < 4; ++i) iv[i] = load32le(chain + 4 * i);
    return with_host(K_PARALLEL, bytes, "belt ECB / CBC blocks", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[2];
        code = s.need(2 * bytes);
        if (code != ERR_OK) return code;
        octet *d = (octet *)s.p;
        B2H_TRY(h2d(d, buf, bytes));
        code = launch_belt_modes(mode, d, d + bytes, nblocks, key, iv, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(buf, d + bytes, bytes));
        return ERR_OK;
    }, [&] { hostp::modes_blocks(hostT(), mode, buf, nblocks, key, iv); });
}

struct belt_ecb_st {          // belt_ecb.c:42-46
    u32 key[8];
    octet block[16];
};
extern "C" size_t beltECB_keep(void) { return sizeof(belt_ecb_st); }
extern "C" void beltECBStart(void *state, const octet key[], size_t len)
try {
    beltKeyExpand2(((belt_ecb_st *)state)->key, key, len);
} B2H_CATCH_VOID("beltECBStart")

static void ecb_step(void *buf_, size_t count, belt_ecb_st *st, int decr)
{
    octet *buf = (octet *)buf_;
    const size_t full = count / 16, tail = count % 16;
    die_on(modes_host(decr ? 1 : 0, buf, full, st->key, nullptr), decr ? "beltECBStepD" : "beltECBStepE");
    if (tail) {
        // ciphertext stealing (belt_ecb.c:74-83,97-106): data shuffling on the host, the block on the GPU
        octet *p = buf + full * 16;
        memcpy(st->block, p, tail);
        memcpy(st->block + tail, p - 16 + tail, 16 - tail);
        if (decr) beltBlockDecr(st->block, st->key); else beltBlockEncr(st->block, st->key);
        memcpy(p, p - 16, tail);
        memcpy(p - 16, st->block, 16);
    }
}
extern "C" void beltECBStepE(void *buf, size_t count, void *state) try { ecb_step(buf, count, (belt_ecb_st *)state, 0); } B2H_CATCH_VOID("beltECBStepE")
extern "C" void beltECBStepD(void *buf, size_t count, void *state) try { ecb_step(buf, count, (belt_ecb_st *)state, 1); } B2H_CATCH_VOID("beltECBStepD")

static err_t ecb_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len, int decr)
{
    if (count < 16 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key) return ERR_BAD_INPUT;
    belt_ecb_st st;
    beltECBStart(&st, key, len);
    memmove(dest, src, count);
    ecb_step(dest, count, &st, decr);
    return ERR_OK;
}
extern "C" err_t beltECBEncr(void *dest, const void *src, size_t count, const octet key[], size_t len)
try {
    return ecb_oneshot(dest, src, count, key, len, 0);
} B2H_CATCH
extern "C" err_t beltECBDecr(void *dest, const void *src, size_t count, const octet key[], size_t len)
try {
    return ecb_oneshot(dest, src, count, key, len, 1);
} B2H_CATCH

// ------------------------------------------------------------------ belt-dwp ---
struct belt_dwp_st {          // belt_dwp.c:27-37 (own layout: no beltPolyMul stack)
    belt_ctr_st ctr;
    u32 r[4];
    u32 t[4];
    uint64_t bits_open, bits_crit;
    octet block[16];
    size_t filled;
};
extern "C" size_t beltDWP_keep(void) { return sizeof(belt_dwp_st); }
extern "C" void beltDWPStart(void *state, const octet key[], size_t len, const octet iv[16])
try {
    belt_dwp_st *st = (belt_dwp_st *)state;
    beltCTRStart(&st->ctr, key, len, iv);                       // ctr = E_K(iv)
    for (int i = 0; i < 4; ++i) st->r[i] = st->ctr.ctr[i];
    beltBlockEncr2(st->r, st->ctr.key);                         // r = E_K(ctr)   (belt_dwp.c:52-54)
    const octet *H = beltH();
    for (int i = 0; i < 4; ++i) st->t[i] = load32le(H + 4 * i); // t = H[0..16)   (:59)
    st->bits_open = st->bits_crit = 0;
    st->filled = 0;
} B2H_CATCH_VOID("beltDWPStart")
extern "C" void beltDWPStepE(void *buf, size_t count, void *state) try { beltCTRStepE(buf, count, &((belt_dwp_st *)state)->ctr); } B2H_CATCH_VOID("beltDWPStepE")
extern "C" void beltDWPStepD(void *buf, size_t count, void *state) try { beltCTRStepE(buf, count, &((belt_dwp_st *)state)->ctr); } B2H_CATCH_VOID("beltDWPStepD")

extern "C" err_t bee2hip_beltDWP_absorb_dev(const void *d_data, size_t nbytes, const u32 r[4], const u32 t[4],
                                            void *d_t_out, void *stream)
try {
    if (misaligned(d_data, 16) || misaligned(d_t_out, 4)) return ERR_BAD_INPUT;
    if ((nbytes && !d_data) || !r || !t || !d_t_out) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_polyhash(d_data, nbytes, r, t, d_t_out, as_stream(stream));
} B2H_CATCH
// t_out <- t after absorbing `nbytes` of host data (zero-padded to whole blocks), on the GPU
static err_t dwp_absorb_host(u32 t_out[4], const u32 t[4], const u32 r[4], const octet *data, size_t nbytes)
{
    return with_host(K_POLY, nbytes, "belt-dwp authentication", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        const size_t off = (nbytes + 15) & ~(size_t)15;
        code = sc.need(off + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        if (nbytes) B2H_TRY(h2d(d, data, nbytes));
        code = launch_belt_polyhash(d, nbytes, r, t, d + off, nullptr);
        if (code != ERR_OK) return code;
        octet out[16];
        B2H_TRY(d2h(out, d + off, 16));
        for (int i = 0; i < 4; ++i) t_out[i] = load32le(out + 4 * i);
        return ERR_OK;
    }, [&] {
        u32 acc[4] = {t[0], t[1], t[2], t[3]};
        hostp::polyhash(acc, r, data, nbytes);
        for (int i = 0; i < 4; ++i) t_out[i] = acc[i];
    });
}
// buffered absorb shared by StepI / StepA (belt_dwp.c:79-106,128-154): whole blocks go to the GPU in one call
static void dwp_feed(belt_dwp_st *st, const octet *p, size_t count, const char *who)
{
    if (st->filled) {
        size_t take = 16 - st->filled;
        if (take > count) take = count;
        memcpy(st->block + st->filled, p, take);
        st->filled += take; p += take; count -= take;
        if (st->filled < 16) return;
        die_on(dwp_absorb_host(st->t, st->t, st->r, st->block, 16), who);
        st->filled = 0;
    }
    const size_t full = count & ~(size_t)15;
    if (full) die_on(dwp_absorb_host(st->t, st->t, st->r, p, full), who);
    if (count - full) { memcpy(st->block, p + full, count - full); st->filled = count - full; }
}
extern "C" void beltDWPStepI(const void *buf, size_t count, void *state)
try {
    belt_dwp_st *st = (belt_dwp_st *)state;
    st->bits_open += (uint64_t)count * 8;
    dwp_feed(st, (const octet *)buf, count, "beltDWPStepI");
} B2H_CATCH_VOID("beltDWPStepI")
extern "C" void beltDWPStepA(const void *buf, size_t count, void *state)
try {
    belt_dwp_st *st = (belt_dwp_st *)state;
    if (count && st->bits_crit == 0 && st->filled) {            // the open data ends here: pad it (belt_dwp.c:115-122)
        die_on(dwp_absorb_host(st->t, st->t, st->r, st->block, st->filled), "beltDWPStepA");
        st->filled = 0;
    }
    st->bits_crit += (uint64_t)count * 8;
    dwp_feed(st, (const octet *)buf, count, "beltDWPStepA");
} B2H_CATCH_VOID("beltDWPStepA")
// the tag of everything absorbed so far; the state is not disturbed (belt_dwp.c:162-189)
static void dwp_tag(octet mac[8], const belt_dwp_st *st, const char *who)
{
    octet tail[32];
    size_t n = 0;
    if (st->filled) { memset(tail, 0, 16); memcpy(tail, st->block, st->filled); n = 16; }
    for (int i = 0; i < 8; ++i) {
        tail[n + i] = (octet)(st->bits_open >> (8 * i));
        tail[n + 8 + i] = (octet)(st->bits_crit >> (8 * i));
    }
    u32 t1[4];
    die_on(dwp_absorb_host(t1, st->t, st->r, tail, n + 16), who);
    beltBlockEncr2(t1, st->ctr.key);
    octet out[16];
    for (int i = 0; i < 4; ++i) store32le(out + 4 * i, t1[i]);
    memcpy(mac, out, 8);
}
extern "C" void beltDWPStepG(octet mac[8], void *state) try { dwp_tag(mac, (const belt_dwp_st *)state, "beltDWPStepG"); } B2H_CATCH_VOID("beltDWPStepG")
extern "C" bool_t beltDWPStepV(const octet mac[8], void *state)
try {
    octet m[8];
    dwp_tag(m, (const belt_dwp_st *)state, "beltDWPStepV");
    return memcmp(m, mac, 8) == 0;
} B2H_CATCH_FALSE("beltDWPStepV")
extern "C" err_t beltDWPWrap(void *dest, octet mac[8], const void *src1, size_t count1, const void *src2,
                             size_t count2, const octet key[], size_t len, const octet iv[16])
try {
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_dwp_st st;
    beltDWPStart(&st, key, len, iv);
    beltDWPStepI(src2, count2, &st);                            // I before E: src2 may overlap dest (belt_dwp.c:218)
    if (count1) memmove(dest, src1, count1);
    beltDWPStepE(dest, count1, &st);
    beltDWPStepA(dest, count1, &st);
    beltDWPStepG(mac, &st);
    return ERR_OK;
} B2H_CATCH
extern "C" err_t beltDWPUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                               const octet mac[8], const octet key[], size_t len, const octet iv[16])
try {
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_dwp_st st;
    beltDWPStart(&st, key, len, iv);
    beltDWPStepI(src2, count2, &st);
    beltDWPStepA(src1, count1, &st);
    if (!beltDWPStepV(mac, &st)) return ERR_BAD_MAC;            // nothing is decrypted (belt_dwp.c:258-262)
    if (count1) memmove(dest, src1, count1);
    beltDWPStepD(dest, count1, &st);
    return ERR_OK;
} B2H_CATCH

// ----------------------------------------------------------------- belt-hash ---
struct belt_hash_st {         // belt_hash.c:28-36 (own layout: h || s contiguous for the kernel)
    u32 hs[12];               // h[8] || s[4]
    uint64_t bits_lo, bits_hi;
    octet block[32];
    size_t filled;
};
extern "C" size_t beltHash_keep(void) { return sizeof(belt_hash_st); }
extern "C" void beltHashStart(void *state)
try {
    belt_hash_st *st = (belt_hash_st *)state;
    const octet *H = beltH();
    for (int i = 0; i < 8; ++i) st->hs[i] = load32le(H + 4 * i);     // h = H[0..32)  (belt_hash.c:52)
    for (int i = 8; i < 12; ++i) st->hs[i] = 0;
    st->bits_lo = st->bits_hi = 0;
    st->filled = 0;
} B2H_CATCH_VOID("beltHashStart")
// hs <- hs after nblocks 32-byte blocks of host data (+ the final length block when fin); on the GPU
static err_t hash_stream_host(u32 hs[12], const octet *data, size_t nblocks, int fin, uint64_t lo, uint64_t hi)
{
    return with_host(K_SERIAL, nblocks * 32, "beltHash", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        const size_t bytes = nblocks * 32;
        code = sc.need(bytes + 64, true);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        if (bytes) B2H_TRY(h2d(d, data, bytes));
        B2H_TRY(h2d(d + bytes, hs, 48));
        code = launch_belt_hash_stream(d + bytes, d, nblocks, fin, lo, hi, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(hs, d + bytes, 48));
        return ERR_OK;
    }, [&] { hostp::hash_stream(hostT(), hs, data, nblocks, fin, lo, hi); });
}
extern "C" void beltHashStepH(const void *buf, size_t count, void *state)
try {
    belt_hash_st *st = (belt_hash_st *)state;
    const octet *p = (const octet *)buf;
    const uint64_t add = (uint64_t)count << 3;                        // 128-bit bit counter (belt_lcl.c:25-51)
    st->bits_lo += add;
    st->bits_hi += ((uint64_t)count >> 61) + (st->bits_lo < add);
    if (st->filled) {
        size_t take = 32 - st->filled;
        if (take > count) take = count;
        memcpy(st->block + st->filled, p, take);
        st->filled += take; p += take; count -= take;
        if (st->filled < 32) return;
        die_on(hash_stream_host(st->hs, st->block, 1, 0, 0, 0), "beltHashStepH");
        st->filled = 0;
    }
    const size_t full = count / 32;
    if (full) die_on(hash_stream_host(st->hs, p, full, 0, 0, 0), "beltHashStepH");
    if (count % 32) { memcpy(st->block, p + 32 * full, count % 32); st->filled = count % 32; }
} B2H_CATCH_VOID("beltHashStepH")
static void hash_digest(octet out[32], const belt_hash_st *st, const char *who)
{
    u32 hs[12];
    memcpy(hs, st->hs, sizeof hs);                                    // the state is not disturbed (belt_hash.c:108-135)
    octet tail[32];
    size_t n = 0;
    if (st->filled) { memset(tail, 0, 32); memcpy(tail, st->block, st->filled); n = 1; }
    die_on(hash_stream_host(hs, tail, n, 1, st->bits_lo, st->bits_hi), who);
    for (int i = 0; i < 8; ++i) store32le(out + 4 * i, hs[i]);
}
extern "C" void beltHashStepG(octet hash[32], void *state) try { hash_digest(hash, (const belt_hash_st *)state, "beltHashStepG"); } B2H_CATCH_VOID("beltHashStepG")
extern "C" void beltHashStepG2(octet hash[], size_t hash_len, void *state)
try {
    octet d[32];
    hash_digest(d, (const belt_hash_st *)state, "beltHashStepG2");
    memcpy(hash, d, hash_len < 32 ? hash_len : 32);
} B2H_CATCH_VOID("beltHashStepG2")
extern "C" bool_t beltHashStepV(const octet hash[32], void *state)
try {
    octet d[32];
    hash_digest(d, (const belt_hash_st *)state, "beltHashStepV");
    return memcmp(d, hash, 32) == 0;
} B2H_CATCH_FALSE("beltHashStepV")
extern "C" bool_t beltHashStepV2(const octet hash[], size_t hash_len, void *state)
try {
    octet d[32];
    hash_digest(d, (const belt_hash_st *)state, "beltHashStepV2");
    return memcmp(d, hash, hash_len < 32 ? hash_len : 32) == 0;
} B2H_CATCH_FALSE("beltHashStepV2")
extern "C" err_t beltHash(octet hash[32], const void *src, size_t count)
try {
    if (!hash || (count && !src)) return ERR_BAD_INPUT;              // belt_hash.c:177-179
    belt_hash_st st;
    beltHashStart(&st);
    beltHashStepH(src, count, &st);
    beltHashStepG(hash, &st);
    return ERR_OK;
} B2H_CATCH

// ------------------------------------------------------------------ belt-sde ---
struct belt_wbl_st {          // belt_lcl.h:143-149
    u32 key[8];
    octet block[16];
    octet sum[16];
    uint64_t round;           // `word` on this ABI
};
struct belt_sde_st {          // belt_sde.c:26-30
    belt_wbl_st wbl[1];
    octet s[16];
};
extern "C" size_t beltSDE_keep(void) { return sizeof(belt_sde_st); }
extern "C" void beltSDEStart(void *state, const octet key[], size_t len)
try {
    belt_sde_st *st = (belt_sde_st *)state;
    beltKeyExpand2(st->wbl->key, key, len);
    st->wbl->round = 0;
} B2H_CATCH_VOID("beltSDEStart")
extern "C" err_t bee2hip_beltSDE_sectors_dev(int decr, void *d_sectors, size_t sector_bytes, size_t nsectors,
                                             const u32 key[8], const void *d_ivs, void *stream)
try {
    if (misaligned(d_sectors, 16) || misaligned(d_ivs, 16)) return ERR_BAD_INPUT;
    if ((decr != 0 && decr != 1) || !key || (nsectors && (!d_sectors || !d_ivs))) return ERR_BAD_INPUT;
    if (sector_bytes % 16 != 0 || sector_bytes < 32) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_sde(decr, d_sectors, sector_bytes / 16, nsectors, key, d_ivs, as_stream(stream));
} B2H_CATCH
static err_t sde_host(int decr, octet *buf, size_t count, const octet iv[16], belt_sde_st *st)
{
    const err_t rc = with_host(K_SERIAL, count, "beltSDE", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        code = sc.need(count + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        B2H_TRY(h2d(d, buf, count));
        B2H_TRY(h2d(d + count, iv, 16));
        code = launch_belt_sde(decr, d, count / 16, 1, st->wbl->key, d + count, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(buf, d, count));
        return ERR_OK;
    }, [&] { hostp::sde_sector(hostT(), decr, buf, count, iv, st->wbl->key); });
    if (rc == ERR_OK) st->wbl->round = decr ? 0 : 2 * (uint64_t)(count / 16);     // where the reference's loops stop (belt_wbl.c)
    return rc;
}
extern "C" void beltSDEStepE(void *buf, size_t count, const octet iv[16], void *state)
try {
    die_on(sde_host(0, (octet *)buf, count, iv, (belt_sde_st *)state), "beltSDEStepE");
} B2H_CATCH_VOID("beltSDEStepE")
extern "C" void beltSDEStepD(void *buf, size_t count, const octet iv[16], void *state)
try {
    die_on(sde_host(1, (octet *)buf, count, iv, (belt_sde_st *)state), "beltSDEStepD");
} B2H_CATCH_VOID("beltSDEStepD")
static err_t sde_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16], int decr)
{
    // belt_sde.c:79-86
    if (count % 16 != 0 || count < 32 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key || !iv)
        return ERR_BAD_INPUT;
    belt_sde_st st;
    beltSDEStart(&st, key, len);
    memmove(dest, src, count);
    return sde_host(decr, (octet *)dest, count, iv, &st);
}
extern "C" err_t beltSDEEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
try {
    return sde_oneshot(dest, src, count, key, len, iv, 0);
} B2H_CATCH
extern "C" err_t beltSDEDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
try {
    return sde_oneshot(dest, src, count, key, len, iv, 1);
} B2H_CATCH

// ------------------------------------------------------------------ belt-che ---
struct belt_che_st {          // belt_che.c:27-41 (own layout).  mac.ctr.key = K, mac.r = E_K(iv); mac.ctr's
    belt_dwp_st mac;          // counter fields are unused
    u32 s[4];
    octet gamma[16];
    size_t reserved;
};
extern "C" size_t beltCHE_keep(void) { return sizeof(belt_che_st); }
extern "C" void beltCHEStart(void *state, const octet key[], size_t len, const octet iv[16])
try {
    belt_che_st *st = (belt_che_st *)state;
    memset(st, 0, sizeof *st);
    beltKeyExpand2(st->mac.ctr.key, key, len);
    for (int i = 0; i < 4; ++i) st->mac.r[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->mac.r, st->mac.ctr.key);                 // r = E_K(iv)  (belt_che.c:54-56)
    for (int i = 0; i < 4; ++i) st->s[i] = st->mac.r[i];        // s = r
    const octet *H = beltH();
    for (int i = 0; i < 4; ++i) st->mac.t[i] = load32le(H + 4 * i);
} B2H_CATCH_VOID("beltCHEStart")
extern "C" err_t bee2hip_beltCHE_blocks_dev(const void *d_src, void *d_dst, size_t nblocks, const u32 key[8],
                                            const u32 s[4], uint64_t first_block, void *d_s_out, void *stream)
try {
    if (misaligned(d_src, 16) || misaligned(d_dst, 16)) return ERR_BAD_INPUT;
    if ((nblocks && (!d_src || !d_dst)) || !key || !s) return ERR_BAD_INPUT;
    if (first_block + nblocks < first_block || first_block + nblocks == ~(uint64_t)0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_che(d_src, d_dst, nblocks, key, s, first_block, d_s_out, as_stream(stream));
} B2H_CATCH
static err_t che_blocks_host(octet *buf, size_t nblocks, belt_che_st *st)
{
    if (nblocks == 0) return ERR_OK;
    const size_t bytes = nblocks * 16;
    return with_host(K_PARALLEL, bytes, "beltCHEStepE", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        code = sc.need(bytes + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        B2H_TRY(h2d(d, buf, bytes));
        code = launch_belt_che(d, d, nblocks, st->mac.ctr.key, st->s, 0, d + bytes, nullptr);
        if (code != ERR_OK) return code;
        octet snew[16];
        B2H_TRY(d2h(snew, d + bytes, 16));
        B2H_TRY(d2h(buf, d, bytes));
        for (int i = 0; i < 4; ++i) st->s[i] = load32le(snew + 4 * i);
        return ERR_OK;
    }, [&] { hostp::che_blocks(hostT(), buf, nblocks, st->mac.ctr.key, st->s); });
}
extern "C" void beltCHEStepE(void *buf_, size_t count, void *state)
try {
    belt_che_st *st = (belt_che_st *)state;
    octet *buf = (octet *)buf_;
    if (st->reserved) {                                         // gamma left from the previous call (belt_che.c:69-83)
        const size_t take = st->reserved < count ? st->reserved : count;
        for (size_t i = 0; i < take; ++i) buf[i] ^= st->gamma[16 - st->reserved + i];
        st->reserved -= take; buf += take; count -= take;
    }
    die_on(che_blocks_host(buf, count / 16, st), "beltCHEStepE");
    buf += count / 16 * 16;
    count %= 16;
    if (count) {                                                // partial block: advance s (bookkeeping, like the CTR
        const u32 out = st->s[3] >> 31;                         // counter increment), gamma = E_K(s) on the GPU
        for (int i = 3; i > 0; --i) st->s[i] = (st->s[i] << 1) | (st->s[i - 1] >> 31);
        st->s[0] = (st->s[0] << 1) ^ (out ? 0x87u : 0u) ^ 1u;
        u32 g[4] = {st->s[0], st->s[1], st->s[2], st->s[3]};
        beltBlockEncr2(g, st->mac.ctr.key);
        for (int i = 0; i < 4; ++i) store32le(st->gamma + 4 * i, g[i]);
        for (size_t i = 0; i < count; ++i) buf[i] ^= st->gamma[i];
        st->reserved = 16 - count;
    }
} B2H_CATCH_VOID("beltCHEStepE")
extern "C" void beltCHEStepD(void *buf, size_t count, void *state) try { beltCHEStepE(buf, count, state); } B2H_CATCH_VOID("beltCHEStepD")
extern "C" void beltCHEStepI(const void *buf, size_t count, void *state) try { beltDWPStepI(buf, count, &((belt_che_st *)state)->mac); } B2H_CATCH_VOID("beltCHEStepI")
extern "C" void beltCHEStepA(const void *buf, size_t count, void *state) try { beltDWPStepA(buf, count, &((belt_che_st *)state)->mac); } B2H_CATCH_VOID("beltCHEStepA")
extern "C" void beltCHEStepG(octet mac[8], void *state) try { dwp_tag(mac, &((const belt_che_st *)state)->mac, "beltCHEStepG"); } B2H_CATCH_VOID("beltCHEStepG")
extern "C" bool_t beltCHEStepV(const octet mac[8], void *state)
try {
    octet m[8];
    dwp_tag(m, &((const belt_che_st *)state)->mac, "beltCHEStepV");
    return memcmp(m, mac, 8) == 0;
} B2H_CATCH_FALSE("beltCHEStepV")
extern "C" err_t beltCHEWrap(void *dest, octet mac[8], const void *src1, size_t count1, const void *src2,
                             size_t count2, const octet key[], size_t len, const octet iv[16])
try {
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_che_st st;
    beltCHEStart(&st, key, len, iv);
    beltCHEStepI(src2, count2, &st);
    if (count1) memmove(dest, src1, count1);
    beltCHEStepE(dest, count1, &st);
    beltCHEStepA(dest, count1, &st);
    beltCHEStepG(mac, &st);
    return ERR_OK;
} B2H_CATCH
extern "C" err_t beltCHEUnwrap(void *dest, const void *src1, size_t count1, const void *src2, size_t count2,
                               const octet mac[8], const octet key[], size_t len, const octet iv[16])
try {
    if ((len != 16 && len != 24 && len != 32) || (count1 && (!src1 || !dest)) || (count2 && !src2) || !key || !iv || !mac)
        return ERR_BAD_INPUT;
    belt_che_st st;
    beltCHEStart(&st, key, len, iv);
    beltCHEStepI(src2, count2, &st);
    beltCHEStepA(src1, count1, &st);
    if (!beltCHEStepV(mac, &st)) return ERR_BAD_MAC;
    if (count1) memmove(dest, src1, count1);
    beltCHEStepD(dest, count1, &st);
    return ERR_OK;
} B2H_CATCH

// ------------------------------------------------------------------ belt-bde ---
struct belt_bde_st {          // belt_bde.c:26-32
    u32 key[8];
    u32 s[4];
    octet block[16];
    octet block1[16];
};
extern "C" size_t beltBDE_keep(void) { return sizeof(belt_bde_st); }
extern "C" void beltBDEStart(void *state, const octet key[], size_t len, const octet iv[16])
try {
    belt_bde_st *st = (belt_bde_st *)state;
    beltKeyExpand2(st->key, key, len);
    for (int i = 0; i < 4; ++i) st->s[i] = load32le(iv + 4 * i);
    beltBlockEncr2(st->s, st->key);                 // s = E_K(iv), on the GPU
} B2H_CATCH_VOID("beltBDEStart")

extern "C" err_t bee2hip_beltBDE_blocks_dev(int decr, const void *d_src, void *d_dst, size_t nblocks,
                                            const u32 key[8], const u32 s[4], uint64_t first_block,
                                            void *d_s_out, void *stream)
try {
    if (misaligned(d_src, 16) || misaligned(d_dst, 16)) return ERR_BAD_INPUT;
    if ((nblocks && (!d_src || !d_dst)) || !key || !s || (decr != 0 && decr != 1)) return ERR_BAD_INPUT;
    if (first_block + nblocks < first_block || first_block + nblocks == ~(uint64_t)0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_bde(decr, d_src, d_dst, nblocks, key, s, first_block, d_s_out, as_stream(stream));
} B2H_CATCH

// whole blocks of a host buffer; advances st->s exactly as the reference's loop does
static err_t bde_host(int decr, octet *buf, size_t nblocks, belt_bde_st *st)
{
    if (nblocks == 0) return ERR_OK;
    const size_t bytes = nblocks * 16;
    const err_t rc = with_host(K_PARALLEL, bytes, "beltBDE", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &sc = t_scr[2];
        code = sc.need(bytes + 16);
        if (code != ERR_OK) return code;
        octet *d = (octet *)sc.p;
        B2H_TRY(h2d(d, buf, bytes));
        code = launch_belt_bde(decr, d, d, nblocks, st->key, st->s, 0, d + bytes, nullptr);
        if (code != ERR_OK) return code;
        octet snew[16];
        B2H_TRY(d2h(snew, d + bytes, 16));
        B2H_TRY(d2h(buf, d, bytes));
        for (int i = 0; i < 4; ++i) st->s[i] = load32le(snew + 4 * i);
        return ERR_OK;
    }, [&] { hostp::bde_blocks(hostT(), decr, buf, nblocks, st->key, st->s); });
    if (rc != ERR_OK) return rc;
    // what the reference's last iteration leaves behind (belt_bde.c:56-63): s, block = <s>, block1 = Y ^ <s>
    octet snew[16];
    for (int i = 0; i < 4; ++i) store32le(snew + 4 * i, st->s[i]);
    memcpy(st->block, snew, 16);
    for (int i = 0; i < 16; ++i) st->block1[i] = buf[bytes - 16 + i] ^ snew[i];
    return ERR_OK;
}
extern "C" void beltBDEStepE(void *buf, size_t count, void *state)
try {
    die_on(bde_host(0, (octet *)buf, count / 16, (belt_bde_st *)state), "beltBDEStepE");
} B2H_CATCH_VOID("beltBDEStepE")
extern "C" void beltBDEStepD(void *buf, size_t count, void *state)
try {
    die_on(bde_host(1, (octet *)buf, count / 16, (belt_bde_st *)state), "beltBDEStepD");
} B2H_CATCH_VOID("beltBDEStepD")
static err_t bde_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16], int decr)
{
    // belt_bde.c:93-100, 118-125
    if (count % 16 != 0 || count < 16 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key || !iv)
        return ERR_BAD_INPUT;
    belt_bde_st st;
    beltBDEStart(&st, key, len, iv);
    memmove(dest, src, count);
    return bde_host(decr, (octet *)dest, count / 16, &st);
}
extern "C" err_t beltBDEEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
try {
    return bde_oneshot(dest, src, count, key, len, iv, 0);
} B2H_CATCH
extern "C" err_t beltBDEDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
try {
    return bde_oneshot(dest, src, count, key, len, iv, 1);
} B2H_CATCH

struct belt_cbc_st {          // belt_cbc.c:63-68
    u32 key[8];
    octet block[16];
    octet block1[16];
};
extern "C" size_t beltCBC_keep(void) { return sizeof(belt_cbc_st); }
extern "C" void beltCBCStart(void *state, const octet key[], size_t len, const octet iv[16])
try {
    belt_cbc_st *st = (belt_cbc_st *)state;
    beltKeyExpand2(st->key, key, len);
    memcpy(st->block, iv, 16);
} B2H_CATCH_VOID("beltCBCStart")

extern "C" void beltCBCStepE(void *buf_, size_t count, void *state)
try {
    belt_cbc_st *st = (belt_cbc_st *)state;
    octet *buf = (octet *)buf_;
    const size_t full = count / 16, tail = count % 16;
    if (full) {
        // the serial chain runs on one lane of the per-message kernel (n = 1), or on the host
        die_on(with_host(K_SERIAL, full * 16, "beltCBCStepE", [&]() -> err_t {
            err_t code = ensure_device();
            if (code != ERR_OK) return code;
            Scratch &s = t_scr[2];
            code = s.need(full * 16 + 16);
            if (code != ERR_OK) return code;
            octet *d = (octet *)s.p;
            B2H_TRY(h2d(d, buf, full * 16));
            B2H_TRY(h2d(d + full * 16, st->block, 16));
            code = launch_belt_cbc_encr(d, full, 1, st->key, d + full * 16, nullptr);
            if (code != ERR_OK) return code;
            octet chain[16];
            B2H_TRY(d2h(chain, d + full * 16, 16));
            B2H_TRY(d2h(buf, d, full * 16));
            memcpy(st->block, chain, 16);
            return ERR_OK;
        }, [&] { hostp::cbc_encr_blocks(hostT(), buf, full, st->key, st->block); }), "beltCBCStepE");
    }
    if (tail) {                                   // stealing, belt_cbc.c:86-93
        octet *p = buf + full * 16;
        for (size_t i = 0; i < tail; ++i) st->block1[i] = p[i] ^ st->block[i];
        memcpy(st->block1 + tail, p - 16 + tail, 16 - tail);
        beltBlockEncr(st->block1, st->key);
        memcpy(p, p - 16, tail);
        memcpy(p - 16, st->block1, 16);
    }
} B2H_CATCH_VOID("beltCBCStepE")

extern "C" void beltCBCStepD(void *buf_, size_t count, void *state)
try {
    belt_cbc_st *st = (belt_cbc_st *)state;
    octet *buf = (octet *)buf_;
    // whole blocks handled by the parallel kernel: all of them, or all but the last full one
    // when a partial tail follows (belt_cbc.c:101-116: "while (count >= 32 || count == 16)")
    const size_t tail = count % 16;
    const size_t par = tail ? count / 16 - 1 : count / 16;
    if (par) {
        octet last[16];
        memcpy(last, buf + (par - 1) * 16, 16);               // becomes the next chaining value
        die_on(modes_host(2, buf, par, st->key, st->block), "beltCBCStepD");
        memcpy(st->block, last, 16);
    }
    if (tail) {                                   // 16 < rest < 32, belt_cbc.c:118-130
        octet *p = buf + par * 16;
        const size_t r = tail;
        memcpy(st->block1, p, 16);
        beltBlockDecr(st->block1, st->key);
        for (size_t i = 0; i < r; ++i) { octet x = st->block1[i]; st->block1[i] = p[16 + i]; p[16 + i] = x; }
        for (size_t i = 0; i < r; ++i) p[16 + i] ^= st->block1[i];
        beltBlockDecr(st->block1, st->key);
        for (int i = 0; i < 16; ++i) p[i] = st->block1[i] ^ st->block[i];
    }
} B2H_CATCH_VOID("beltCBCStepD")

static err_t cbc_oneshot(void *dest, const void *src, size_t count, const octet key[], size_t len,
                         const octet iv[16], int decr)
{
    if (count < 16 || (len != 16 && len != 24 && len != 32) || !src || !dest || !key || !iv) return ERR_BAD_INPUT;
    belt_cbc_st st;
    beltCBCStart(&st, key, len, iv);
    memmove(dest, src, count);
    if (decr) beltCBCStepD(dest, count, &st); else beltCBCStepE(dest, count, &st);
    return ERR_OK;
}
extern "C" err_t beltCBCEncr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
try {
    return cbc_oneshot(dest, src, count, key, len, iv, 0);
} B2H_CATCH
extern "C" err_t beltCBCDecr(void *dest, const void *src, size_t count, const octet key[], size_t len,
                             const octet iv[16])
try {
    return cbc_oneshot(dest, src, count, key, len, iv, 1);
} B2H_CATCH

