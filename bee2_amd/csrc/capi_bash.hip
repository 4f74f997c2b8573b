// capi_bash.hip -- bash-f and the bash hashing drop-ins (bash.h:127-225).  Part of the C ABI (capi.hip).
// =================================================================== bash ====
extern "C" const char bash_platform[] = "BASH_HIP_GFX950";

extern "C" void bashF(octet block[192], void *stack)
try {
    (void)stack;                                   // bashF_deep() == 0
    die_on(with_host(K_PRIM, 192, "bashF", [&] { return bee2hip_bashF_batch(block, 1); }, [&] { hostp::bashF(block); }), "bashF");
} B2H_CATCH_VOID("bashF")
extern "C" size_t bashF_deep(void) { return 0; }


// ============================================================= bash hashing ===
// bash_hash_st / belt_mac_st (bee2 layouts) are defined in mixed_kernels.hip
extern "C" size_t bashHash_keep(void) { return sizeof(bash_hash_st); }   // + bashF_deep() == 0

extern "C" void bashHashStart(void *state, size_t l)
try {
    bash_hash_st *st = (bash_hash_st *)state;
    memset(st->s, 0, sizeof st->s);
    st->s[192 - 8] = (octet)(l / 4);
    st->buf_len = 192 - l / 2;
    st->pos = 0;
} B2H_CATCH_VOID("bashHashStart")

// run the device sponge over `count` host bytes for one state
static err_t sponge_gpu(bash_hash_st *st, const octet *buf, size_t count)
{
    Scratch &s = t_scr[0];
    err_t code = s.need(sizeof(bash_hash_st) + count + 16, true);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, st, sizeof *st));
    B2H_TRY(h2d(d + sizeof *st, buf, count));
    const octet *dd = d + sizeof *st;
    // large chunk: byte-wise up to the next block boundary, whole rate blocks with 8 lanes (a 3x shorter chain,
    // DESIGN.md 4.7), the remainder byte-wise again
    const size_t head = st->pos ? st->buf_len - st->pos : 0;
    if (count >= 4096 + head) {
        const size_t blocks = (count - head) / st->buf_len, tail = count - head - blocks * st->buf_len;
        if (head) { code = launch_bash_sponge(d, dd, 0, head, 1, 0, nullptr); if (code != ERR_OK) return code; }
        code = launch_bash_sponge_cols(d, dd + head, blocks, nullptr);
        if (code != ERR_OK) return code;
        if (tail) { code = launch_bash_sponge(d, dd + head + blocks * st->buf_len, 0, tail, 1, 0, nullptr); if (code != ERR_OK) return code; }
    } else {
        code = launch_bash_sponge(d, dd, 0, count, 1, 0, nullptr);
        if (code != ERR_OK) return code;
    }
    B2H_TRY(d2h(st, d, sizeof *st));
    return ERR_OK;
}

static err_t sponge_host(bash_hash_st *st, const octet *buf, size_t count)
{
    return with_host(K_SERIAL, count, "bashHashStepH", [&] { return sponge_gpu(st, buf, count); },
                     [&] { hostp::sponge_absorb(st->s, st->buf_len, &st->pos, buf, count); });
}

extern "C" void bashHashStepH(const void *buf, size_t count, void *state)
try {
    bash_hash_st *st = (bash_hash_st *)state;
    // not a full rate block yet: buffering only, no permutation (bash_hash.c:57-62)
    if (count < st->buf_len - st->pos) {
        memcpy(st->s + st->pos, buf, count);
        st->pos += count;
        return;
    }
    die_on(sponge_host(st, (const octet *)buf, count), "bashHashStepH");
} B2H_CATCH_VOID("bashHashStepH")

static void hash_final(bash_hash_st *st)
{
    // s1 = s, pad with 0x40 0.. (bash_hash.c:86-100), one more bashF -- on the GPU
    memcpy(st->s1, st->s, 192);
    memset(st->s1 + st->pos, 0, st->buf_len - st->pos);
    st->s1[st->pos] = 0x40;
    die_on(with_host(K_PRIM, 192, "bashHashStepG", [&] { return bee2hip_bashF_batch(st->s1, 1); }, [&] { hostp::bashF(st->s1); }),
           "bashHashStepG");
}

extern "C" void bashHashStepG(octet hash[], size_t hash_len, void *state)
try {
    bash_hash_st *st = (bash_hash_st *)state;
    hash_final(st);
    memmove(hash, st->s1, hash_len);
} B2H_CATCH_VOID("bashHashStepG")

extern "C" bool_t bashHashStepV(const octet hash[], size_t hash_len, void *state)
try {
    bash_hash_st *st = (bash_hash_st *)state;
    hash_final(st);
    return memcmp(hash, st->s1, hash_len) == 0;
} B2H_CATCH_FALSE("bashHashStepV")

extern "C" err_t bashHash(octet hash[], size_t l, const void *src, size_t count)
try {
    if (l == 0 || l % 16 != 0 || l > 256) return ERR_BAD_PARAMS;
    if ((count && !src) || !hash) return ERR_BAD_INPUT;
    bash_hash_st *st = new (std::nothrow) bash_hash_st;
    if (!st) return ERR_OUTOFMEMORY;
    bashHashStart(st, l);
    bashHashStepH(src, count, st);
    bashHashStepG(hash, l / 4, st);
    delete st;
    return ERR_OK;
} B2H_CATCH

