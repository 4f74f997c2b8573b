// capi_base.hip -- management entry points, the device- and host-pointer batch entries of the primitives (bashF, belt block,
// CTR blocks) and the helpers every drop-in file uses (encr_host_blocks, die_on).  Part of the C ABI (capi.hip).
// ============================================================== management ===
extern "C" err_t bee2hip_set_device(int device)
try {
    B2H_TRY(hipSetDevice(device));
    return ensure_device();
} B2H_CATCH
extern "C" err_t bee2hip_sync(void *stream)
try {
    B2H_TRY(hipStreamSynchronize(as_stream(stream)));
    return ERR_OK;
} B2H_CATCH
extern "C" const char *bee2hip_last_error(void) { return t_err; }
extern "C" const char *bee2hip_version(void) { return "bee2hip 0.1 gfx950"; }

// ===================================================== device-pointer batch ===
extern "C" err_t bee2hip_bashF_batch_dev(void *d_states, size_t n, void *stream)
try {
    if (misaligned(d_states, 16)) return ERR_BAD_INPUT;
    if (n && !d_states) return ERR_BAD_INPUT;
    return launch_bashF_batch(d_states, n, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_beltCTR_blocks_dev(void *d_buf, size_t nblocks, const u32 key[8],
                                            const u32 ctr0[4], uint64_t first_block, void *stream)
try {
    if (misaligned(d_buf, 16)) return ERR_BAD_INPUT;
    if ((nblocks && !d_buf) || !key || !ctr0) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_ctr_blocks(d_buf, nblocks, key, ctr0, first_block, nullptr, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_beltBlockEncr_dev(void *d_blocks, size_t nblocks, const u32 key[8], void *stream)
try {
    if (misaligned(d_blocks, 16)) return ERR_BAD_INPUT;
    if ((nblocks && !d_blocks) || !key) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_belt_encr_blocks(d_blocks, nblocks, key, as_stream(stream));
} B2H_CATCH

// ======================================================= host-pointer batch ===
extern "C" err_t bee2hip_bashF_batch(octet *states, size_t n)
try {
    if (n == 0) return ERR_OK;
    if (!states) return ERR_BAD_INPUT;
    Scratch &s = t_scr[0];
    err_t code = s.need(n * 192);
    if (code != ERR_OK) return code;
    if (n * 192 >= DUPLEX_MIN)              // chunks of 2^16 states = 12 MiB: upload, permute and download overlap
        return duplex_inplace(states, (octet *)s.p, 192, n, (size_t)1 << g_duplex_log2_states,
                              [](octet *d, size_t, size_t cnt, hipStream_t st) { return launch_bashF_batch(d, cnt, st); });
    B2H_TRY(h2d(s.p, states, n * 192));
    code = launch_bashF_batch(s.p, n, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(states, s.p, n * 192));
    return ERR_OK;
} B2H_CATCH

// E_K over host blocks (n small): the only way the drop-in layer evaluates belt
static err_t encr_host_blocks(uint32_t *blocks, size_t n, const u32 key[8])
{
    return with_host(K_PRIM, n * 16, "belt block encryption", [&]() -> err_t {
        err_t code = ensure_device();
        if (code != ERR_OK) return code;
        Scratch &s = t_scr[1];
        code = s.need(n * 16);
        if (code != ERR_OK) return code;
        B2H_TRY(h2d(s.p, blocks, n * 16));
        code = launch_belt_encr_blocks(s.p, n, key, nullptr);
        if (code != ERR_OK) return code;
        B2H_TRY(d2h(blocks, s.p, n * 16));
        return ERR_OK;
    }, [&] { for (size_t i = 0; i < n; ++i) hostp::belt_encr(hostT(), blocks + 4 * i, key); });
}

// a device failure inside a void bee2 function cannot be reported through the bee2
// signature: fail loudly instead of returning wrong bytes.
static void die_on(err_t code, const char *where)
{
    if (code == ERR_OK) return;
    fprintf(stderr, "libbee2hip: %s failed (err %u): %s\n", where, (unsigned)code, t_err);
    abort();
}

