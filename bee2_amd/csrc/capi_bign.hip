// capi_bign.hip -- bign: parameter sets, verification (general, one signer, keyed), public-key validation, key generation,
// signing; batch entries and bee2 drop-ins (bign.h, bign128/192/256.h).  Part of the C ABI (capi.hip).
// ==================================================================== bign ===
// STB 34.101.45 annex B parameter sets: k_bign{128,192,256}_{p,a,b,q,yG,seed} come from
// bign_curves.inc (generated from the reference's bignParamsStd, bign_params.c:34-230)
// DER of the pre-hash OIDs the level-fixed facades use (bign128.c:151-153, bign192.c:151-153,
// bign256.c:151-153): belt-hash, bash384, bash512
static const octet k_oid_belt_hash[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};
static const octet k_oid_bash384[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x4D, 0x0C};
static const octet k_oid_bash512[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x4D, 0x0D};

struct StdCurve { size_t l; const char *name; const octet *p, *a, *b, *q, *yG, *seed; };
static const StdCurve k_curves[3] = {
    {128, "1.2.112.0.2.0.34.101.45.3.1", k_bign128_p, k_bign128_a, k_bign128_b, k_bign128_q, k_bign128_yG, k_bign128_seed},
    {192, "1.2.112.0.2.0.34.101.45.3.2", k_bign192_p, k_bign192_a, k_bign192_b, k_bign192_q, k_bign192_yG, k_bign192_seed},
    {256, "1.2.112.0.2.0.34.101.45.3.3", k_bign256_p, k_bign256_a, k_bign256_b, k_bign256_q, k_bign256_yG, k_bign256_seed},
};

extern "C" err_t bignParamsStd(bign_params *params, const char *name)
try {
    if (!params || !name) return ERR_BAD_INPUT;
    memset(params, 0, sizeof *params);
    for (const StdCurve &c : k_curves) {
        if (strcmp(name, c.name) == 0) {
            const size_t no = c.l / 4;
            params->l = c.l;
            memcpy(params->p, c.p, no);
            memcpy(params->a, c.a, no);
            memcpy(params->seed, c.seed, 8);
            memcpy(params->b, c.b, no);
            memcpy(params->q, c.q, no);
            memcpy(params->yG, c.yG, no);
            return ERR_OK;
        }
    }
    return ERR_FILE_NOT_FOUND;
} B2H_CATCH

static bool all_zero(const octet *p, size_t n)
{
    octet acc = 0;
    for (size_t i = 0; i < n; ++i) acc |= p[i];
    return acc == 0;
}

// bignParamsCheck (bign_params.c:244-280).  *standard = one of the three parameter sets of STB 34.101.45 annex B, which
// have their own kernels; anything else that passes goes to the general-curve kernels where they exist (verification,
// public-key validation) and is ERR_NOT_IMPLEMENTED elsewhere (the constant-time signing path).
static err_t params_check2(const bign_params *params, bool *standard)
{
    *standard = false;
    if (!params) return ERR_BAD_INPUT;
    if (2 * params->l % 64) return ERR_NOT_IMPLEMENTED;
    const size_t no = 2 * params->l / 8;
    if (no == 0 || no > 64) return ERR_BAD_PARAMS;
    const bool ok = params->p[0] % 4 == 3 && params->q[0] % 2 == 1 && params->p[no - 1] >= 128 &&
                    params->q[no - 1] >= 128 && all_zero(params->p + no, 64 - no) &&
                    !all_zero(params->a, no) && !all_zero(params->b, no) &&
                    all_zero(params->a + no, 64 - no) && all_zero(params->b + no, 64 - no) &&
                    all_zero(params->q + no, 64 - no) && all_zero(params->yG + no, 64 - no);
    if (!ok) return ERR_BAD_PARAMS;
    if (params->l % 64) return ERR_NOT_IMPLEMENTED;
    if (params->l != 128 && params->l != 192 && params->l != 256) return ERR_BAD_PARAMS;
    for (const StdCurve &c : k_curves) {
        if (c.l != params->l) continue;
        *standard = !(memcmp(params->p, c.p, no) || memcmp(params->a, c.a, no) || memcmp(params->b, c.b, no) ||
                      memcmp(params->q, c.q, no) || memcmp(params->yG, c.yG, no));
        break;
    }
    return ERR_OK;
}
// for the entry points that serve the standard curves only
// the signing side (bignPubkeyCalc, bignKeypairGen, bignSign*): every set bignParamsCheck + bignEcCreate accept;
// *standard tells which kernels serve it (the table-driven ones of bign_sign_kernels.hip, or the general-curve
// constant-time ladder of bign_generic_kernels.hip)
static err_t params_check_sign(const bign_params *params, bool *standard)
{
    const err_t code = params_check2(params, standard);
    if (code != ERR_OK) return code;
    return *standard ? ERR_OK : bign_generic_check(params);
}
static err_t params_check(const bign_params *params)
{
    bool standard;
    return params_check_sign(params, &standard);
}
static err_t pubkey_calc_any(const bign_params *params, bool keygen, const void *d_privkeys, size_t n, void *d_pubkeys, void *d_codes)
{
    bool standard;
    const err_t code = params_check_sign(params, &standard);
    if (code != ERR_OK) return code;
    return standard ? launch_bign_pubkey_calc(params->l, keygen, d_privkeys, n, d_pubkeys, d_codes, nullptr)
                    : launch_bign_pubkey_calc_generic(params, keygen, d_privkeys, n, d_pubkeys, d_codes, nullptr);
}

// oidFromDER(0, der, count) != SIZE_MAX  (src/core/oid.c:94-101, src/core/der.c:114-258,921-975):
// tag 0x06, definite minimal length covering the whole buffer, sub-identifiers without a
// leading 0x80 octet and below 2^32.
static bool oid_der_valid(const octet *der, size_t count)
{
    if (!der || count < 2 || count == (size_t)-1) return false;
    if (der[0] != 0x06) return false;
    size_t len, hdr;
    if (der[1] < 128) { len = der[1]; hdr = 2; }
    else {
        const size_t r = der[1] - 128;
        if (der[1] == 128 || der[1] == 255 || r > sizeof(size_t) || count < 2 + r) return false;
        if (der[2] == 0 || (r == 1 && der[2] < 128)) return false;
        len = 0;
        for (size_t i = 0; i < r; ++i) len = (len << 8) | der[2 + i];
        hdr = 2 + r;
    }
    if (hdr + len != count) return false;
    const octet *v = der + hdr;
    u32 val = 0;
    for (size_t pos = 0; pos < len; ++pos) {
        if (val & 0xFE000000u) return false;
        if (val == 0 && v[pos] == 128) return false;
        val = (val << 7) | (v[pos] & 127u);
        if ((v[pos] & 128) == 0) val = 0;
    }
    return true;
}

extern "C" err_t bee2hip_bignVerifyL_batch_dev(size_t l, const octet oid_der[], size_t oid_len,
                                               const void *d_hashes, const void *d_sigs,
                                               const void *d_pubkeys, size_t n, void *d_codes,
                                               void *stream)
try {
    if (misaligned(d_hashes, 16) || misaligned(d_sigs, 16) || misaligned(d_pubkeys, 16) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_sigs || !d_pubkeys || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_verify(l, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_bignVerify_batch_dev(const octet oid_der[], size_t oid_len,
                                              const void *d_hashes, const void *d_sigs,
                                              const void *d_pubkeys, size_t n, void *d_codes,
                                              void *stream)
try {
    return bee2hip_bignVerifyL_batch_dev(128, oid_der, oid_len, d_hashes, d_sigs, d_pubkeys, n, d_codes, stream);
} B2H_CATCH

extern "C" err_t bee2hip_bign128Verify_batch_dev(const void *d_hashes, const void *d_sigs,
                                                 const void *d_pubkeys, size_t n, void *d_codes,
                                                 void *stream)
try {
    return bee2hip_bignVerifyL_batch_dev(128, k_oid_belt_hash, sizeof k_oid_belt_hash, d_hashes, d_sigs,
                                         d_pubkeys, n, d_codes, stream);
} B2H_CATCH

extern "C" err_t bee2hip_bignVerify_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                          const octet *hashes, const octet *sigs, const octet *pubkeys,
                                          size_t n, err_t *codes)
try {
    // order of checks as bignVerify: params first (bign_sign.c:355-356), then inputs, then OID
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {                              // bignEcCreate judges the parameters next (bign_sign.c:357-358)
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!hashes || !sigs || !pubkeys || !codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;                 // octets per field element
    const size_t hb = no * n, sb = (no + no / 2) * n, pb = 2 * no * n;
    const size_t so = (hb + 15) & ~(size_t)15, po = (so + sb + 15) & ~(size_t)15, co = (po + pb + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (standard && n >= VERIFY_PIPE_MIN && g_verify_pipe) {
        // big host batch: chunk c + 1 is uploaded (its own stream) while the kernels of chunk c run -- the 144 n octets
        // of input are a quarter of the time of the whole call otherwise (profiles/r03_verify_hostpipe.txt)
        code = t_duplex.get();
        if (code != ERR_OK) return code;
        const hipStream_t up = t_duplex.up, st = t_duplex.dn;
        const size_t CH = VERIFY_PIPE_CHUNK, nch = (n + CH - 1) / CH, sg = no + no / 2;
        std::vector<hipEvent_t> ev(nch, nullptr);
        hipError_t he = hipSuccess;
        for (size_t c = 0; c < nch && he == hipSuccess; ++c) he = hipEventCreateWithFlags(&ev[c], hipEventDisableTiming);
        for (size_t c = 0; c < nch && he == hipSuccess && code == ERR_OK; ++c) {
            const size_t first = c * CH, cnt = std::min(CH, n - first);
            he = hipMemcpyAsync(d + first * no, hashes + first * no, cnt * no, hipMemcpyHostToDevice, up);
            if (he == hipSuccess) he = hipMemcpyAsync(d + so + first * sg, sigs + first * sg, cnt * sg, hipMemcpyHostToDevice, up);
            if (he == hipSuccess) he = hipMemcpyAsync(d + po + first * 2 * no, pubkeys + first * 2 * no, cnt * 2 * no, hipMemcpyHostToDevice, up);
            if (he == hipSuccess) he = hipEventRecord(ev[c], up);
            if (he == hipSuccess) he = hipStreamWaitEvent(st, ev[c], 0);
            if (he == hipSuccess)
                code = launch_bign_verify(params->l, oid_der, oid_len, d + first * no, d + so + first * sg, d + po + first * 2 * no,
                                          cnt, d + co + 4 * first, st);
        }
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        (void)hipStreamSynchronize(up);
        for (size_t c = 0; c < nch; ++c) if (ev[c]) (void)hipEventDestroy(ev[c]);
        if (code != ERR_OK) return code;
        B2H_TRY(he);
        B2H_TRY(hipMemcpy(codes, d + co, 4 * n, hipMemcpyDeviceToHost));
        return ERR_OK;
    }
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + so, sigs, sb));
    B2H_TRY(h2d(d + po, pubkeys, pb));
    code = standard ? launch_bign_verify(params->l, oid_der, oid_len, d, d + so, d + po, n, d + co, nullptr)
                    : launch_bign_verify_generic(params, oid_der, oid_len, d, d + so, d + po, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
} B2H_CATCH

// the three standard curves for host_bign.hpp: c, q, y_G and the fixed tables of G, built at first use
template <int N>
static const hostb::Curve<N> &host_curve(int which, uint64_t c)
{
    static hostb::Curve<N> E;
    static std::once_flag once;
    std::call_once(once, [&] { E.init(c, k_curves[which].q, k_curves[which].yG); });
    return E;
}
static err_t verify_one_host(size_t l, const octet oid_der[], size_t oid_len, const octet hash[], const octet sig[],
                             const octet pubkey[])
{
    const hostp::BeltTables &T = hostT();
    if (l == 128) return hostb::verify<4>(host_curve<4>(0, BIGN128_CRANDALL_C), T, host_beltH(), oid_der, oid_len, hash, sig, pubkey);
    if (l == 192) return hostb::verify<6>(host_curve<6>(1, BIGN192_CRANDALL_C), T, host_beltH(), oid_der, oid_len, hash, sig, pubkey);
    return hostb::verify<8>(host_curve<8>(2, BIGN256_CRANDALL_C), T, host_beltH(), oid_der, oid_len, hash, sig, pubkey);
}

// ONE key pair / public key / signature on a standard curve, on the calling core in constant-time arithmetic
// (host_bign_ct.hpp; its header says what that covers).  BEE2HIP_FORCE=gpu keeps every secret in the GPU kernels.
template <int N>
static const hostct::SignCurve<N> &host_sign_curve(int which, uint64_t c)
{
    static hostct::SignCurve<N> S;
    static std::once_flag once;
    std::call_once(once, [&] { S.init(host_curve<N>(which, c)); });
    return S;
}
static bool sign_on_host(const bign_params *params)
{
    bool standard;
    return params_check2(params, &standard) == ERR_OK && standard && host_wanted(K_SIGN1, 1);
}
// -> an error code of bee2, or ERR_OUTOFMEMORY when the window table could not be built
static err_t pubkey_calc_one_host(size_t l, bool keygen, const octet *privkey, octet *pubkey)
{
    if (l == 128) { const auto &S = host_sign_curve<4>(0, BIGN128_CRANDALL_C); return S.ready ? hostct::pubkey_calc<4>(S, keygen, privkey, pubkey) : ERR_OUTOFMEMORY; }
    if (l == 192) { const auto &S = host_sign_curve<6>(1, BIGN192_CRANDALL_C); return S.ready ? hostct::pubkey_calc<6>(S, keygen, privkey, pubkey) : ERR_OUTOFMEMORY; }
    const auto &S = host_sign_curve<8>(2, BIGN256_CRANDALL_C);
    return S.ready ? hostct::pubkey_calc<8>(S, keygen, privkey, pubkey) : ERR_OUTOFMEMORY;
}
static err_t sign_one_host(size_t l, const octet oid_der[], size_t oid_len, const octet *hash, const octet *privkey, const octet *k,
                           const void *t, size_t t_len, octet *sig)
{
    const hostp::BeltTables &T = hostT();
    const octet *H = host_beltH();
    if (l == 128) { const auto &S = host_sign_curve<4>(0, BIGN128_CRANDALL_C); return S.ready ? hostct::sign<4>(S, T, H, oid_der, oid_len, hash, privkey, k, t, t_len, sig) : ERR_OUTOFMEMORY; }
    if (l == 192) { const auto &S = host_sign_curve<6>(1, BIGN192_CRANDALL_C); return S.ready ? hostct::sign<6>(S, T, H, oid_der, oid_len, hash, privkey, k, t, t_len, sig) : ERR_OUTOFMEMORY; }
    const auto &S = host_sign_curve<8>(2, BIGN256_CRANDALL_C);
    return S.ready ? hostct::sign<8>(S, T, H, oid_der, oid_len, hash, privkey, k, t, t_len, sig) : ERR_OUTOFMEMORY;
}

// ---- n signatures under ONE public key (bign_kernels.hip "one signer") ----
// device-resident hashes / signatures, the key on the host; standard curve
static err_t verify_onekey_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes, const void *d_sigs,
                               const octet pubkey[], size_t n, void *d_codes, hipStream_t st)
{
    const size_t no = l / 4;
    err_t code = launch_bign_verify_onekey(l, oid_der, oid_len, d_hashes, d_sigs, pubkey, n, d_codes, st);
    if (code != ERR_KEY_NOT_ON_CURVE) return code;
    // a key off the curve (or with a coordinate >= p): bee2 does not check (bign_sign.c:306-311), and the comb table of such a
    // point proves nothing about the reference's walk -- the general path with the key n times gives the reference's codes
    void *rep = nullptr;
    code = scratch_for_stream(st, 7, 2 * no * (n + 1), &rep);
    if (code != ERR_OK) return code;
    octet *d_key = (octet *)rep + 2 * no * n;
    B2H_TRY(hipMemcpyAsync(d_key, pubkey, 2 * no, hipMemcpyHostToDevice, st));
    B2H_TRY(hipStreamSynchronize(st));                 // (pubkey is the caller's, pageable)
    code = launch_replicate_key(d_key, 2 * no, n, rep, st);
    if (code != ERR_OK) return code;
    return launch_bign_verify(l, oid_der, oid_len, d_hashes, d_sigs, rep, n, d_codes, st);
}

extern "C" err_t bee2hip_bignVerifyL_onekey_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                                      const void *d_sigs, const octet pubkey[], size_t n, void *d_codes, void *stream)
try {
    if (misaligned(d_hashes, 16) || misaligned(d_sigs, 16) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_sigs || !pubkey || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return verify_onekey_dev(l, oid_der, oid_len, d_hashes, d_sigs, pubkey, n, d_codes, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_bignVerify_onekey_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                 const octet *hashes, const octet *sigs, const octet pubkey[], size_t n, err_t *codes)
try {
    // order of checks as bignVerify / bee2hip_bignVerify_batch
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!hashes || !sigs || !pubkey || !codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    if (!standard) {                                   // general-curve kernels: the key n times through the batch entry
        try {
            std::vector<octet> rep(2 * no * n);
            for (size_t i = 0; i < n; ++i) memcpy(rep.data() + 2 * no * i, pubkey, 2 * no);
            return bee2hip_bignVerify_batch(params, oid_der, oid_len, hashes, sigs, rep.data(), n, codes);
        } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
    }
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t hb = no * n, sb = (no + no / 2) * n;
    const size_t so = (hb + 15) & ~(size_t)15, co = (so + sb + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + so, sigs, sb));
    code = verify_onekey_dev(params->l, oid_der, oid_len, d, d + so, pubkey, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
} B2H_CATCH

// ---- n signatures of K signers: key_index[i] < nkeys says whose signature i is ----
extern "C" err_t bee2hip_bignVerifyL_keyed_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                                     const void *d_sigs, const octet pubkeys[], size_t nkeys,
                                                     const void *d_key_index, size_t n, void *d_codes, void *stream)
try {
    if (misaligned(d_hashes, 16) || misaligned(d_sigs, 16) || misaligned(d_codes, 4) || misaligned(d_key_index, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_sigs || !pubkeys || !nkeys || !d_key_index || !d_codes)) return ERR_BAD_INPUT;
    if (nkeys > 4096) return ERR_BAD_INPUT;             // (more signers than that: the general entry)
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_verify_keyed(l, oid_der, oid_len, d_hashes, d_sigs, pubkeys, nkeys, d_key_index, n, d_codes, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_bignVerify_keyed_batch(const bign_params *params, const octet oid_der[], size_t oid_len,
                                                const octet *hashes, const octet *sigs, const octet *pubkeys, size_t nkeys,
                                                const u32 *key_index, size_t n, err_t *codes)
try {
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!hashes || !sigs || !pubkeys || !nkeys || !key_index || !codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    const size_t no = params->l / 4;
    if (!standard || nkeys > 4096) {                   // general-curve kernels / a crowd of signers: every signature with its key, the general entry
        try {
            std::vector<octet> rep(2 * no * n);
            std::vector<size_t> bad;
            for (size_t i = 0; i < n; ++i) {
                if (key_index[i] >= nkeys) { bad.push_back(i); memcpy(rep.data() + 2 * no * i, pubkeys, 2 * no); }
                else memcpy(rep.data() + 2 * no * i, pubkeys + 2 * no * key_index[i], 2 * no);
            }
            code = bee2hip_bignVerify_batch(params, oid_der, oid_len, hashes, sigs, rep.data(), n, codes);
            if (code == ERR_OK) for (size_t i : bad) codes[i] = ERR_BAD_INPUT;
            return code;
        } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
    }
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t hb = no * n, sb = (no + no / 2) * n;
    const size_t so = (hb + 15) & ~(size_t)15, io = (so + sb + 15) & ~(size_t)15, co = (io + 4 * n + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + so, sigs, sb));
    B2H_TRY(h2d(d + io, key_index, 4 * n));
    code = launch_bign_verify_keyed(params->l, oid_der, oid_len, d, d + so, pubkeys, nkeys, d + io, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
} B2H_CATCH

extern "C" err_t bignVerify(const bign_params *params, const octet oid_der[], size_t oid_len,
                            const octet hash[], const octet sig[], const octet pubkey[])
try {
    err_t one = ERR_BAD_SIG;
    bool standard;
    err_t pc = params_check2(params, &standard);
    if (!hash || !sig || !pubkey) {
        if (pc == ERR_OK && !standard) pc = bign_generic_check(params);
        return pc != ERR_OK ? pc : ERR_BAD_INPUT;
    }
    // ONE signature on a standard curve: the calling core (host_bign.hpp) unless BEE2HIP_FORCE=gpu; same order of checks
    // as the batch entry (parameters, inputs, OID), same requirement of a usable device
    if (pc == ERR_OK && standard && host_wanted(K_VERIFY1, 1) && oid_der_valid(oid_der, oid_len)) {
        const err_t code = device_seen();
        if (code != ERR_OK) return code;
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        return verify_one_host(params->l, oid_der, oid_len, hash, sig, pubkey);
    }
    const err_t code = bee2hip_bignVerify_batch(params, oid_der, oid_len, hash, sig, pubkey, 1, &one);
    if (code == ERR_BEE2HIP_DEVICE && pc == ERR_OK && standard && force_mode() != FORCE_GPU) {
        // a device fault under a single verification: finished on the host like the void drop-ins (with_host)
        fprintf(stderr, "libbee2hip: bignVerify: device path failed (%s); finished on the host\n", t_err);
        g_n_fallback.fetch_add(1, std::memory_order_relaxed);
        return verify_one_host(params->l, oid_der, oid_len, hash, sig, pubkey);
    }
    return code != ERR_OK ? code : one;
} B2H_CATCH

static err_t level_verify(int which, const octet *oid, const octet *hash, const octet *sig, const octet *pubkey)
{
    bign_params params;
    bignParamsStd(&params, k_curves[which].name);
    return bignVerify(&params, oid, 11, hash, sig, pubkey);
}
extern "C" err_t bign128Verify(const octet hash[32], const octet sig[48], const octet pubkey[64])
try {
    return level_verify(0, k_oid_belt_hash, hash, sig, pubkey);
} B2H_CATCH
extern "C" err_t bign192Verify(const octet hash[48], const octet sig[72], const octet pubkey[96])
try {
    return level_verify(1, k_oid_bash384, hash, sig, pubkey);
} B2H_CATCH
extern "C" err_t bign256Verify(const octet hash[64], const octet sig[96], const octet pubkey[128])
try {
    return level_verify(2, k_oid_bash512, hash, sig, pubkey);
} B2H_CATCH

// ---- public-key validation (bign_misc.c:319-365) ----
extern "C" err_t bee2hip_bignPubkeyValL_batch_dev(size_t l, const void *d_pubkeys, size_t n, void *d_codes,
                                                  void *stream)
try {
    if (misaligned(d_pubkeys, 16) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_pubkeys || !d_codes)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_pubkey_val(l, d_pubkeys, n, d_codes, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_bignPubkeyVal_batch(const bign_params *params, const octet *pubkeys, size_t n,
                                             err_t *codes)
try {
    // bignPubkeyVal: params first (bign_misc.c:358-361), then the key
    bool standard;
    err_t code = params_check2(params, &standard);
    if (code != ERR_OK) return code;
    if (!standard) {
        code = bign_generic_check(params);
        if (code != ERR_OK) return code;
    }
    if (n && (!pubkeys || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t pb = params->l / 2 * n, co = (pb + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + n * 4);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    B2H_TRY(h2d(d, pubkeys, pb));
    code = standard ? launch_bign_pubkey_val(params->l, d, n, d + co, nullptr)
                    : launch_bign_pubkey_val_generic(params, d, n, d + co, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(codes, d + co, 4 * n));
    return ERR_OK;
} B2H_CATCH

extern "C" err_t bignPubkeyVal(const bign_params *params, const octet pubkey[])
try {
    err_t one = ERR_BAD_PUBKEY;
    if (!pubkey) {
        bool standard;
        err_t pc = params_check2(params, &standard);
        if (pc == ERR_OK && !standard) pc = bign_generic_check(params);
        return pc != ERR_OK ? pc : ERR_BAD_INPUT;
    }
    {   // ONE key on a standard curve: the calling core (host_bign.hpp) unless BEE2HIP_FORCE=gpu
        bool standard;
        if (params_check2(params, &standard) == ERR_OK && standard && host_wanted(K_VERIFY1, 1)) {
            const err_t code = device_seen();
            if (code != ERR_OK) return code;
            g_n_host.fetch_add(1, std::memory_order_relaxed);
            const size_t l = params->l;
            if (l == 128) return hostb::pubkey_val<4>(host_curve<4>(0, BIGN128_CRANDALL_C), k_curves[0].b, pubkey);
            if (l == 192) return hostb::pubkey_val<6>(host_curve<6>(1, BIGN192_CRANDALL_C), k_curves[1].b, pubkey);
            return hostb::pubkey_val<8>(host_curve<8>(2, BIGN256_CRANDALL_C), k_curves[2].b, pubkey);
        }
    }
    const err_t code = bee2hip_bignPubkeyVal_batch(params, pubkey, 1, &one);
    return code != ERR_OK ? code : one;
} B2H_CATCH
static err_t level_pubkey_val(int which, const octet *pubkey)
{
    bign_params params;
    bignParamsStd(&params, k_curves[which].name);
    return bignPubkeyVal(&params, pubkey);
}
extern "C" err_t bign128PubkeyVal(const octet pubkey[64]) try { return level_pubkey_val(0, pubkey); } B2H_CATCH
extern "C" err_t bign192PubkeyVal(const octet pubkey[96]) try { return level_pubkey_val(1, pubkey); } B2H_CATCH
extern "C" err_t bign256PubkeyVal(const octet pubkey[128]) try { return level_pubkey_val(2, pubkey); } B2H_CATCH

// ---- 8f-4 tail: public key from private key, key generation, signing (bign_misc.c:182-229,373-417,
// bign_sign.c:32-245).  Secrets cross the staging buffer t_scr[3]; it is overwritten with zeros before return.
static void wipe_dev(void *p, size_t n) { if (p && n) (void)zero_staging(p, n); }
// staged secrets are zeroed on EVERY way out of a host entry point (early error returns, an allocation that throws)
struct WipeGuard {
    void *p;
    size_t n;
    ~WipeGuard() { wipe_dev(p, n); }
};
// host bytes that hold secrets: wiped on every way out of the scope, a return in mid-function and an exception included
struct SecretVec {
    std::vector<octet> v;
    ~SecretVec() { if (!v.empty()) wipe_host(v.data(), v.size()); }
};

extern "C" err_t bee2hip_bignPubkeyCalcL_batch_dev(size_t l, const void *d_privkeys, size_t n, void *d_pubkeys,
                                                   void *d_codes, void *stream)
try {
    if (misaligned(d_privkeys, 4) || misaligned(d_pubkeys, 4) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_privkeys || !d_pubkeys || !d_codes)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_pubkey_calc(l, false, d_privkeys, n, d_pubkeys, d_codes, as_stream(stream));
} B2H_CATCH
extern "C" err_t bee2hip_bignSign2L_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                              const void *d_privkeys, const void *d_t, size_t t_len, int t_shared,
                                              size_t n, void *d_sigs, void *d_codes, void *stream)
try {
    if (misaligned(d_hashes, 16) || misaligned(d_privkeys, 4) || misaligned(d_sigs, 4) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_privkeys || !d_sigs || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (!d_t) t_len = 0;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_sign(l, 0, oid_der, oid_len, d_hashes, d_privkeys, t_len ? d_t : nullptr, t_len, t_shared, n, d_sigs,
                            d_codes, as_stream(stream));
} B2H_CATCH
extern "C" err_t bee2hip_bignSignKL_batch_dev(size_t l, const octet oid_der[], size_t oid_len, const void *d_hashes,
                                              const void *d_privkeys, const void *d_ks, size_t n, void *d_sigs,
                                              void *d_codes, void *stream)
try {
    if (misaligned(d_hashes, 16) || misaligned(d_privkeys, 4) || misaligned(d_ks, 4) || misaligned(d_sigs, 4) || misaligned(d_codes, 4)) return ERR_BAD_INPUT;
    if (l != 128 && l != 192 && l != 256) return ERR_BAD_PARAMS;
    if (n && (!d_hashes || !d_privkeys || !d_ks || !d_sigs || !d_codes)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_bign_sign(l, 1, oid_der, oid_len, d_hashes, d_privkeys, d_ks, 0, 0, n, d_sigs, d_codes, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_bignPubkeyCalc_batch(const bign_params *params, const octet *privkeys, size_t n,
                                              octet *pubkeys, err_t *codes)
try {
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    if (n && (!privkeys || !pubkeys || !codes)) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;
    const size_t db = no * n, po = (db + 15) & ~(size_t)15, co = (po + 2 * no * n + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(co + 4 * n);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    const WipeGuard wipe{d, db};
    B2H_TRY(h2d(d, privkeys, db));
    code = pubkey_calc_any(params, false, d, n, d + po, d + co);
    if (code == ERR_OK) {
        hipError_t e = d2h(codes, d + co, 4 * n);
        // bee2 leaves the output alone when it fails: copy the keys of the good items only
        octet *tmp = new (std::nothrow) octet[2 * no * n];
        if (!tmp) return ERR_OUTOFMEMORY;                       // (the guard above wipes the staged keys)
        if (e == hipSuccess) e = d2h(tmp, d + po, 2 * no * n);
        if (e != hipSuccess) code = hip_fail(e, "bignPubkeyCalc copy");
        else for (size_t i = 0; i < n; ++i) if (codes[i] == ERR_OK) memcpy(pubkeys + 2 * no * i, tmp + 2 * no * i, 2 * no);
        delete[] tmp;
    }
    return code;
} B2H_CATCH

// mode 0: t (shared by the batch, may be null) -- bignSign2; mode 1: aux = one-time keys k[n][no] -- bignSign after its rng
static err_t sign_batch_host(int mode, const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                             const octet *privkeys, const octet *aux, size_t t_len, size_t n, octet *sigs, err_t *codes)
{
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    if (n && (!hashes || !privkeys || !sigs || !codes || (mode == 1 && !aux))) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (n == 0) return ERR_OK;
    if (mode == 0 && !aux) t_len = 0;
    code = ensure_device();
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4, sg = no + no / 2;
    // theta = belt-hash(oid || d || t) for additional input beyond what the nonce kernel assembles itself (bign_sign.c:183-196): ONE
    // ragged belt-hash launch over the n messages (round 5, ADVICE r04: the streaming drop-ins took ~4 synchronous GPU round trips
    // per signature -- tens of seconds at 2^18 -- and could only abort on a device fault).  The messages hold the private keys: the
    // host copy lives in a vector that wipes itself on every way out, the device copy under a WipeGuard.
    SecretVec theta;
    int dev_mode = mode;
    size_t ab = mode == 1 ? no * n : t_len;
    if (mode == 0 && t_len > 64) {
        const size_t ml = oid_len + no + t_len;
        const size_t o_off = (n * ml + 15) & ~(size_t)15, o_th = (o_off + 8 * (n + 1) + 15) & ~(size_t)15;
        SecretVec msgs;
        msgs.v.resize(n * ml);
        for (size_t i = 0; i < n; ++i) {
            octet *m = msgs.v.data() + i * ml;
            memcpy(m, oid_der, oid_len);
            memcpy(m + oid_len, privkeys + no * i, no);
            memcpy(m + oid_len + no, aux, t_len);
        }
        std::vector<uint64_t> off(n + 1);
        for (size_t i = 0; i <= n; ++i) off[i] = (uint64_t)i * ml;
        Scratch &ts = t_scr[2];
        code = ts.need(o_th + 32 * n + 16);
        if (code != ERR_OK) return code;
        octet *td = (octet *)ts.p;
        const WipeGuard wipe_msgs{td, o_th + 32 * n};
        B2H_TRY(h2d(td, msgs.v.data(), n * ml));
        B2H_TRY(h2d(td + o_off, off.data(), 8 * (n + 1)));
        code = launch_hash_ragged(0, td, td + o_off, nullptr, n, td + o_th, nullptr, true);     // (bank-private table: the messages hold d)
        if (code != ERR_OK) return code;
        theta.v.resize(32 * n);
        B2H_TRY(d2h(theta.v.data(), td + o_th, 32 * n));
        aux = theta.v.data();
        ab = 32 * n;
        dev_mode = 2;
    }
    const size_t hb = no * n;
    const size_t o_d = (hb + 15) & ~(size_t)15, o_a = (o_d + hb + 15) & ~(size_t)15, o_s = (o_a + ab + 15) & ~(size_t)15,
                 o_c = (o_s + sg * n + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    code = s.need(o_c + 4 * n);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    const WipeGuard wipe{d + o_d, o_s - o_d};              // private keys and one-time keys / t / theta
    B2H_TRY(h2d(d, hashes, hb));
    B2H_TRY(h2d(d + o_d, privkeys, hb));
    if (ab) B2H_TRY(h2d(d + o_a, aux, ab));
    {
        bool standard;
        code = params_check_sign(params, &standard);
        if (code == ERR_OK)
            code = standard ? launch_bign_sign(params->l, dev_mode, oid_der, oid_len, d, d + o_d, ab ? d + o_a : nullptr, t_len, 1, n,
                                               d + o_s, d + o_c, nullptr)
                            : launch_bign_sign_generic(params, dev_mode, oid_der, oid_len, d, d + o_d, ab ? d + o_a : nullptr, t_len, 1,
                                                       n, d + o_s, d + o_c, nullptr);
    }
    if (code == ERR_OK) {
        hipError_t e = d2h(codes, d + o_c, 4 * n);
        std::vector<octet> tmp(sg * n);
        if (e == hipSuccess) e = d2h(tmp.data(), d + o_s, sg * n);
        if (e != hipSuccess) code = hip_fail(e, "bignSign copy");
        else for (size_t i = 0; i < n; ++i) if (codes[i] == ERR_OK) memcpy(sigs + sg * i, tmp.data() + sg * i, sg);
    }
    return code;
}
extern "C" err_t bee2hip_bignSign2_batch(const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                                         const octet *privkeys, const void *t, size_t t_len, size_t n, octet *sigs, err_t *codes)
try {
    return sign_batch_host(0, params, oid_der, oid_len, hashes, privkeys, (const octet *)t, t_len, n, sigs, codes);
} B2H_CATCH
extern "C" err_t bee2hip_bignSignK_batch(const bign_params *params, const octet oid_der[], size_t oid_len, const octet *hashes,
                                         const octet *privkeys, const octet *ks, size_t n, octet *sigs, err_t *codes)
try {
    try {
        return sign_batch_host(1, params, oid_der, oid_len, hashes, privkeys, ks, 0, n, sigs, codes);
    } catch (const std::bad_alloc &) { return ERR_OUTOFMEMORY; }
} B2H_CATCH

// ---- drop-ins.  Order of checks as the reference: parameters (bignParamsCheck), pointers, OID, private key.
extern "C" err_t bignPubkeyCalc(octet pubkey[], const bign_params *params, const octet privkey[])
try {
    err_t one = ERR_BAD_PRIVKEY;
    if (!pubkey || !privkey) {
        const err_t pc = params_check(params);
        return pc != ERR_OK ? pc : ERR_BAD_INPUT;
    }
    if (sign_on_host(params)) {
        const err_t dc = device_seen();            // the library still needs its GPU (no GPU-less operation)
        if (dc != ERR_OK) return dc;
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        return pubkey_calc_one_host(params->l, false, privkey, pubkey);
    }
    const err_t code = bee2hip_bignPubkeyCalc_batch(params, privkey, 1, pubkey, &one);
    return code != ERR_OK ? code : one;
} B2H_CATCH
// zzRandNZMod (zz_mod.c:463-485) on the host, exactly as bee2 calls the caller's generator: draws of no octets
// until 0 < a < mod, at most B_PER_IMPOSSIBLE + 1 = 65 of them.  The comparison is the only arithmetic involved.
static bool rand_nz_mod(octet *a, const octet *mod, size_t no, gen_i rng, void *rng_state)
{
    for (int tries = 0; tries <= 64; ++tries) {
        rng(a, no, rng_state);
        bool zero = true, less = false;
        for (size_t i = 0; i < no; ++i) zero = zero && a[i] == 0;
        for (size_t i = no; i-- > 0;) {
            if (a[i] != mod[i]) { less = a[i] < mod[i]; break; }
        }
        if (!zero && less) return true;
    }
    return false;
}
extern "C" err_t bignKeypairGen(octet privkey[], octet pubkey[], const bign_params *params, gen_i rng, void *rng_state)
try {
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    if (!privkey || !pubkey) return ERR_BAD_INPUT;
    if (!rng) return ERR_BAD_RNG;
    const size_t no = params->l / 4;
    octet d[64];
    // bignKeypairGenEc draws d below the FIELD modulus p (bign_misc.c:209), not below q
    if (!rand_nz_mod(d, params->p, no, rng, rng_state)) return ERR_BAD_RNG;
    // any d below 2^(2l) is multiplied, as bignMulBase does (no range check against q here)
    code = ensure_device();
    if (code == ERR_OK && sign_on_host(params)) {
        octet q2[128];
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        code = pubkey_calc_one_host(params->l, true, d, q2);      // ERR_BAD_PARAMS when d G = O (bign_misc.c:214-218)
        if (code == ERR_OK) { memcpy(privkey, d, no); memcpy(pubkey, q2, 2 * no); }
        wipe_host(d, sizeof d);
        return code;
    }
    if (code == ERR_OK) {
        Scratch &s = t_scr[3];
        code = s.need(64 + 128 + 16);
        if (code == ERR_OK) {
            octet *dd = (octet *)s.p;
            hipError_t e = h2d(dd, d, no);
            if (e == hipSuccess) {
                code = pubkey_calc_any(params, true, dd, 1, dd + 64, dd + 192);
                octet q[128];
                err_t one = ERR_BAD_PARAMS;
                if (code == ERR_OK) e = d2h(q, dd + 64, 2 * no);
                if (code == ERR_OK && e == hipSuccess) e = d2h(&one, dd + 192, 4);
                if (code == ERR_OK && e == hipSuccess) {
                    code = one;                                  // ERR_BAD_PARAMS when d G = O (bign_misc.c:214-218)
                    if (one == ERR_OK) { memcpy(privkey, d, no); memcpy(pubkey, q, 2 * no); }
                }
            }
            if (e != hipSuccess) code = hip_fail(e, "bignKeypairGen copy");
            wipe_dev(dd, 64);
        }
    }
    memset(d, 0, sizeof d);
    return code;
} B2H_CATCH
extern "C" err_t bignSign(octet sig[], const bign_params *params, const octet oid_der[], size_t oid_len, const octet hash[],
                          const octet privkey[], gen_i rng, void *rng_state)
try {
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;
    if (!hash || !privkey || !sig || (hash < sig + no + no / 2 && sig < hash + no)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    if (!rng) return ERR_BAD_RNG;
    // d first (bign_sign.c:62-68): a bad key must not consume the generator
    {
        bool zero = true, less = false;
        for (size_t i = 0; i < no; ++i) zero = zero && privkey[i] == 0;
        for (size_t i = no; i-- > 0;) if (privkey[i] != params->q[i]) { less = privkey[i] < params->q[i]; break; }
        if (zero || !less) return ERR_BAD_PRIVKEY;
    }
    octet k[64];
    if (!rand_nz_mod(k, params->q, no, rng, rng_state)) return ERR_BAD_RNG;
    err_t one = ERR_BAD_PRIVKEY;
    if (sign_on_host(params)) {
        code = device_seen();
        if (code == ERR_OK) {
            g_n_host.fetch_add(1, std::memory_order_relaxed);
            octet out[96];
            code = sign_one_host(params->l, oid_der, oid_len, hash, privkey, k, nullptr, 0, out);
            if (code == ERR_OK) memcpy(sig, out, no + no / 2);
        }
        wipe_host(k, sizeof k);
        return code;
    }
    code = bee2hip_bignSignK_batch(params, oid_der, oid_len, hash, privkey, k, 1, sig, &one);
    wipe_host(k, sizeof k);
    return code != ERR_OK ? code : one;
} B2H_CATCH
extern "C" err_t bignSign2(octet sig[], const bign_params *params, const octet oid_der[], size_t oid_len, const octet hash[],
                           const octet privkey[], const void *t, size_t t_len)
try {
    err_t code = params_check(params);
    if (code != ERR_OK) return code;
    const size_t no = params->l / 4;
    if (!hash || !privkey || !sig || (hash < sig + no + no / 2 && sig < hash + no)) return ERR_BAD_INPUT;
    if (!oid_der_valid(oid_der, oid_len)) return ERR_BAD_OID;
    err_t one = ERR_BAD_PRIVKEY;
    if (sign_on_host(params)) {
        code = device_seen();
        if (code != ERR_OK) return code;
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        octet out[96];                                   // sig may alias nothing else, but is written only on success
        code = sign_one_host(params->l, oid_der, oid_len, hash, privkey, nullptr, t, t ? t_len : 0, out);
        if (code == ERR_OK) memcpy(sig, out, no + no / 2);
        return code;
    }
    code = bee2hip_bignSign2_batch(params, oid_der, oid_len, hash, privkey, t, t_len, 1, sig, &one);
    return code != ERR_OK ? code : one;
} B2H_CATCH
#define B2H_LEVEL_FACADE(L, IDX, OID, NO)                                                                          \
    extern "C" err_t bign##L##PubkeyCalc(octet pubkey[2 * NO], const octet privkey[NO])                             \
    try { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignPubkeyCalc(pubkey, &p, privkey); } B2H_CATCH            \
    extern "C" err_t bign##L##KeypairGen(octet privkey[NO], octet pubkey[2 * NO], gen_i rng, void *rng_state)       \
    try { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignKeypairGen(privkey, pubkey, &p, rng, rng_state); } B2H_CATCH \
    extern "C" err_t bign##L##Sign(octet sig[NO + NO / 2], const octet hash[NO], const octet privkey[NO], gen_i rng, void *rng_state) \
    try { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignSign(sig, &p, OID, 11, hash, privkey, rng, rng_state); } B2H_CATCH \
    extern "C" err_t bign##L##Sign2(octet sig[NO + NO / 2], const octet hash[NO], const octet privkey[NO], const void *t, size_t t_len) \
    try { bign_params p; bignParamsStd(&p, k_curves[IDX].name); return bignSign2(sig, &p, OID, 11, hash, privkey, t, t_len); } B2H_CATCH
B2H_LEVEL_FACADE(128, 0, k_oid_belt_hash, 32)
B2H_LEVEL_FACADE(192, 1, k_oid_bash384, 48)
B2H_LEVEL_FACADE(256, 2, k_oid_bash512, 64)
#undef B2H_LEVEL_FACADE

#ifdef BEE2HIP_EXPERIMENTS
extern "C" err_t bee2hip_debug_fe(int op, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream)
try {
    return launch_bign_debug_fe(128, op, d_a, d_b, d_out, n, as_stream(stream));
} B2H_CATCH
extern "C" err_t bee2hip_debug_feL(size_t l, int op, const void *d_a, const void *d_b, void *d_out, size_t n,
                                   void *stream)
try {
    return launch_bign_debug_fe(l, op, d_a, d_b, d_out, n, as_stream(stream));
} B2H_CATCH
#endif

