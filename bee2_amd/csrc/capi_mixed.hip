// capi_mixed.hip -- the per-message bash + belt-MAC batch (configs[4]), the path policy, ragged hash batches (bsum front-end).
// Part of the C ABI (capi.hip).
// ====================================================== mixed batch (H4) ===
extern "C" err_t bee2hip_bashHash_beltMAC_batch_dev(const void *d_msgs, size_t msg_len, size_t n, size_t l,
                                                    const octet key[], size_t key_len,
                                                    void *d_digests, void *d_tags, void *stream)
try {
    if (misaligned(d_msgs, 16)) return ERR_BAD_INPUT;
    const bool do_hash = d_digests != nullptr, do_mac = d_tags != nullptr;
    if (do_hash && (l == 0 || l % 16 != 0 || l > 256)) return ERR_BAD_PARAMS;      // bash_hash.c:122-123
    if (do_mac && ((key_len != 16 && key_len != 24 && key_len != 32) || !key)) return ERR_BAD_INPUT;
    if (n && msg_len && !d_msgs) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    u32 kw[8] = {0};
    if (do_mac) beltKeyExpand2(kw, key, key_len);
    return launch_bashHash_beltMAC(d_msgs, msg_len, n, l, kw, do_hash, do_mac, d_digests, d_tags,
                                   as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_bashHash_beltMAC_batch(const octet *msgs, size_t msg_len, size_t n, size_t l,
                                                const octet key[], size_t key_len,
                                                octet *digests, octet *tags)
try {
    if (digests && (l == 0 || l % 16 != 0 || l > 256)) return ERR_BAD_PARAMS;
    if (tags && ((key_len != 16 && key_len != 24 && key_len != 32) || !key)) return ERR_BAD_INPUT;
    if (n && msg_len && !msgs) return ERR_BAD_INPUT;
    if (n == 0) return ERR_OK;
    const size_t dlen = digests ? l / 4 : 0;
    const size_t in_b = (n * msg_len + 15) & ~(size_t)15, dg_b = (n * dlen + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    err_t code = s.need(in_b + dg_b + n * 8 + 16);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (n * msg_len) B2H_TRY(h2d(d, msgs, n * msg_len));
    code = bee2hip_bashHash_beltMAC_batch_dev(d, msg_len, n, l, key, key_len, digests ? d + in_b : nullptr,
                                              tags ? d + in_b + dg_b : nullptr, nullptr);
    if (code != ERR_OK) return code;
    if (digests) B2H_TRY(d2h(digests, d + in_b, n * dlen));
    if (tags) B2H_TRY(d2h(tags, d + in_b + dg_b, n * 8));
    return ERR_OK;
} B2H_CATCH

// ============================================================ path policy (product ABI) ===
extern "C" int bee2hip_path_policy(int mode)
{
    const int was = bee2hip::force_mode();
    if (mode >= 0) bee2hip::g_force.store(mode == 1 ? bee2hip::FORCE_GPU : mode == 2 ? bee2hip::FORCE_CPU : bee2hip::FORCE_AUTO);
    return was;
}
// drop-in helper calls so far: which = 0 host path (by size or by BEE2HIP_FORCE=cpu), 1 GPU path, 2 finished on the host
// after the GPU path failed twice
extern "C" unsigned long long bee2hip_path_count(int which)
{
    return which == 0 ? bee2hip::g_n_host.load() : which == 1 ? bee2hip::g_n_gpu.load() : bee2hip::g_n_fallback.load();
}


// ================================================= 8f-3: ragged hash batches ===
extern "C" err_t bee2hip_hash_ragged_ordered_dev(size_t alg, const void *d_data, const void *d_offsets,
                                                 const void *d_order, size_t n, void *d_digests, void *stream)
try {
    if (misaligned(d_offsets, 8) || misaligned(d_order, 4) || misaligned(d_digests, 4)) return ERR_BAD_INPUT;
    if (alg != 0 && alg != 128 && alg != 192 && alg != 256) return ERR_BAD_PARAMS;
    if (n && (!d_offsets || !d_digests)) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    return launch_hash_ragged(alg, d_data, d_offsets, d_order, n, d_digests, as_stream(stream));
} B2H_CATCH

extern "C" err_t bee2hip_hash_ragged_dev(size_t alg, const void *d_data, const void *d_offsets, size_t n,
                                         void *d_digests, void *stream)
try {
    return bee2hip_hash_ragged_ordered_dev(alg, d_data, d_offsets, nullptr, n, d_digests, stream);
} B2H_CATCH

static err_t hash_ragged_host(size_t alg, const octet *data, const uint64_t *offsets, size_t n, octet *digests);
extern "C" err_t bee2hip_hash_ragged(size_t alg, const octet *data, const uint64_t *offsets, size_t n,
                                     octet *digests)
try {
    return hash_ragged_host(alg, data, offsets, n, digests);
} B2H_CATCH
static err_t hash_ragged_host(size_t alg, const octet *data, const uint64_t *offsets, size_t n, octet *digests)
{
    if (alg != 0 && alg != 128 && alg != 192 && alg != 256) return ERR_BAD_PARAMS;
    if (n == 0) return ERR_OK;
    if (!offsets || !digests) return ERR_BAD_INPUT;
    for (size_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return ERR_BAD_INPUT;
    const size_t total = (size_t)offsets[n] , dlen = alg ? alg / 4 : 32;
    if (total && !data) return ERR_BAD_INPUT;
    if (n > 0xffffffffull) return ERR_BAD_INPUT;
    // longest first: the 64 lanes of a wavefront then hold messages of similar length and the long
    // serial chains start at once (bench.py "hash_ragged": +20 % belt-hash, +57 % bash256)
    std::vector<uint32_t> ord(n);
    for (size_t i = 0; i < n; ++i) ord[i] = (uint32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [offsets](uint32_t a, uint32_t b) {
        return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b];
    });
    // A message is ONE dependent chain: a GPU lane (pair) walks it at ~0.12-0.15 us per octet, a host core at ~0.008.  When a
    // few messages are far longer than the rest the batch would wait for their chains (a 256 KiB message: 26-30 ms; a 1 GiB
    // file: minutes) with the device otherwise idle, so this HOST-pointer entry -- the data is in host memory anyway --
    // hands the K longest messages to host threads (host_small.hpp, as the drop-in beltHash / bashHash of one message
    // does) while the GPU takes the rest.  K balances the two sides: it grows while the host threads would finish before
    // the GPU's longest remaining chain.  The device-pointer entries never do this.  BEE2HIP_FORCE=gpu: K = 0.
    size_t K = 0;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = std::min<size_t>(hw ? hw : 1, 16);
    if (force_mode() != FORCE_GPU && n >= 2) {
        // seconds per octet of ONE chain: on a host thread / on the GPU (bench.py ragged leg).  A message is one chain on one
        // thread, so the host side's time is max(its longest chain, all its octets / T) -- not the sum / T
        const double c_host = alg ? 5.0e-9 : 8.4e-9, c_gpu = alg ? 4.0e-8 : 1.2e-7;
        const auto len_of = [&](size_t k) { return (double)(offsets[ord[k] + 1] - offsets[ord[k]]); };
        double sum = 0, best = len_of(0) * c_gpu;                          // K = 0: the GPU's longest chain decides
        while (K + 1 < n && len_of(K) >= 65536.0) {                        // chains under ~8 ms are the GPU's
            const double host_s = std::max(len_of(0), (sum + len_of(K)) / (double)T) * c_host;
            const double both = std::max(host_s, len_of(K + 1) * c_gpu);   // K + 1 messages on the host, the rest on the GPU
            if (both >= best) break;
            best = both;
            sum += len_of(K);
            ++K;
        }
    }
    std::vector<octet> hdig(K * dlen);
    std::vector<std::thread> workers;
    std::atomic<size_t> next{0};
    const auto host_job = [&] {
        const hostp::BeltTables &HT = hostT();
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= K) return;
            const size_t i = ord[k];
            const octet *m = data + offsets[i];
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
            octet *out = hdig.data() + k * dlen;
            if (alg == 0) {
                hostb::BeltHashPieces bh(HT, host_beltH());
                bh.absorb(m, len);
                bh.digest(out);
            } else {
                octet st[192];
                memset(st, 0, sizeof st);
                st[192 - 8] = (octet)(alg / 4);                             // bashHashStart (bash_hash.c:38-48)
                const size_t rate = 192 - alg / 2;
                size_t pos = 0;
                hostp::sponge_absorb(st, rate, &pos, m, len);
                memset(st + pos, 0, rate - pos);                            // bashHashStepG (bash_hash.c:84-102)
                st[pos] = 0x40;
                hostp::bashF(st);
                memcpy(out, st, dlen);
            }
        }
    };
    struct Joiner {                                                        // joined on every way out
        std::vector<std::thread> &w;
        ~Joiner() { for (std::thread &t : w) if (t.joinable()) t.join(); }
    } joiner{workers};
    if (K) {
        g_n_host.fetch_add(1, std::memory_order_relaxed);
        for (size_t t = 0; t < std::min(T, K); ++t) workers.emplace_back(host_job);
    }
    const size_t ng = n - K;                                               // slots of the GPU launch: ord[K .. n)
    const size_t ob = (n + 1) * 8, oo = (total + 15) & ~(size_t)15, ro = (oo + ob + 15) & ~(size_t)15,
                 go = (ro + n * 4 + 15) & ~(size_t)15;
    Scratch &s = t_scr[3];
    err_t code = s.need(go + n * dlen + 16);
    if (code != ERR_OK) return code;
    octet *d = (octet *)s.p;
    if (K == 0) {
        if (total) B2H_TRY(h2d(d, data, total));
    } else {
        // the K messages the host threads take are not uploaded: the runs of octets between them go up one by one (the device
        // layout is the caller's, with holes nobody reads)
        std::vector<uint32_t> host_ix(ord.begin(), ord.begin() + K);
        std::sort(host_ix.begin(), host_ix.end());
        uint64_t from = 0;
        for (size_t k = 0; k <= K; ++k) {
            const uint64_t to = k < K ? offsets[host_ix[k]] : (uint64_t)total;
            if (to > from) B2H_TRY(h2d(d + from, data + from, (size_t)(to - from)));
            if (k < K) from = offsets[host_ix[k] + 1];
        }
    }
    B2H_TRY(h2d(d + oo, offsets, ob));
    B2H_TRY(h2d(d + ro, ord.data() + K, ng * 4));
    code = bee2hip_hash_ragged_ordered_dev(alg, d, d + oo, d + ro, ng, d + go, nullptr);
    if (code != ERR_OK) return code;
    B2H_TRY(d2h(digests, d + go, n * dlen));
    for (std::thread &t : workers) t.join();
    for (size_t k = 0; k < K; ++k) memcpy(digests + (size_t)ord[k] * dlen, hdig.data() + k * dlen, dlen);
    return ERR_OK;
}

