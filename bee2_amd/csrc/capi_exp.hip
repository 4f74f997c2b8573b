// capi_exp.hip -- the hooks of include/bee2hip_internal.h: libbee2hip_exp.so only.  Part of the C ABI (capi.hip).
#ifdef BEE2HIP_EXPERIMENTS      // everything from here to the end of the kernel-timing hook: libbee2hip_exp.so only
// ============================================================ internal tuning hook ===
// A/B switch for experiment builds (tools/ab/bashf_ab.py); not part of the product ABI (BEE2HIP_INTERNAL).
namespace bee2hip { void set_bashF_variant(int v); void set_ctr_variant(int v); void set_verify_path(int v); void set_verify_split(int v); void set_sign_coop(int v); void set_sign_wg(int v); void set_fused_tab(int v); void set_long_hash_form(int v); void set_ragged_fork(int v); void set_verify_pairs(int v); void set_onekey_tab16(int v); void set_onekey_slots(int v); void set_onekey_quads(int v); void set_inv_lanes(int v); }
extern "C" err_t bee2hip_internal_tune(int key, int value)
try {
    switch (key) {
    case 0: bee2hip::set_bashF_variant(value); return ERR_OK;
    case 1: bee2hip::set_ctr_variant(value); return ERR_OK;
    case 2: bee2hip::set_verify_path(value); return ERR_OK;
    case 3: bee2hip::g_pinned_limit = value < 0 ? 0 : (size_t)value > bee2hip::PINNED_MAX ? bee2hip::PINNED_MAX : (size_t)value; return ERR_OK;
    case 4: bee2hip::g_force.store(value == 1 ? bee2hip::FORCE_GPU : value == 2 ? bee2hip::FORCE_CPU : bee2hip::FORCE_AUTO); return ERR_OK;   // as BEE2HIP_FORCE
    case 5: bee2hip::g_inject_fail.store(value); return ERR_OK;       // tests: the next `value` GPU attempts of drop-in helpers fail
    case 8: bee2hip::set_verify_split(value); return ERR_OK;          // parts of a big verification batch (0 by size, 1 never, 2..4)
    case 6: bee2hip::g_duplex_log2_states = value; return ERR_OK;     // chunk of the duplex host pipeline, bashF states (log2)
    case 7: bee2hip::g_duplex_log2_blocks = value; return ERR_OK;     //                                   belt blocks (log2)
    case 12: bee2hip::set_sign_wg(value); return ERR_OK;              // largest workgroup of the signing side's hashing kernels
    case 11: bee2hip::g_verify_pipe = value; return ERR_OK;           // chunked upload of big host-pointer verification batches
    case 10: bee2hip::set_sign_coop(value); return ERR_OK;            // lanes per scalar of k G, signing side (0 = by batch size)
    case 9: bee2hip::g_duplex_ramp = value; return ERR_OK;            // ramped chunk sizes at the ends of the pipeline
    case 13: bee2hip::set_fused_tab(value); return ERR_OK;            // belt table of the fused bash + belt-mac kernel (A/B)
    case 16: bee2hip::set_long_hash_form(value); return ERR_OK;        // table / workgroup of the long belt-hash kernel (A/B)
    case 19: bee2hip::set_verify_pairs(value); return ERR_OK;          // verification main kernel: multiply-adds in pairs (-1 by size, 0 never, else always)
    case 23: bee2hip::set_inv_lanes(value); return ERR_OK;             // lanes of the shared-inversion kernel of verification (log2; 0 = by curve)
    case 22: bee2hip::set_onekey_quads(value); return ERR_OK;          // one-signer verification: four lanes per signature (-1 by size, 0 never, 1 always)
    case 21: bee2hip::set_onekey_slots(value); return ERR_OK;          // one-signer verification: keys the table cache keeps (tests: evictions under load)
    case 20: bee2hip::set_onekey_tab16(value); return ERR_OK;          // one-signer verification: log2 of the signatures after which a key gets its 16-bit table (-1 by curve, 63 never)
    case 17: bee2hip::set_ragged_fork(value); return ERR_OK;           // ragged hashing: long chains and short messages on two queues (1) or one (0)
    case 14: bee2hip::g_duplex_fail_chunk.store(value); return ERR_OK;   // tests: the duplex host pipeline fails at this chunk (1-based) ...
    case 15: bee2hip::g_duplex_fail_times.store(value); return ERR_OK;   // ... in the next `value` pipelines
    case 24: bee2hip::g_new_fail_in.store(value); return ERR_OK;       // tests: the value-th operator new of the library from now on throws std::bad_alloc (0 = off)
    default: return ERR_BAD_INPUT;
    }
} B2H_CATCH

// drop-in helper calls so far: which = 0 host path (by size or by BEE2HIP_FORCE=cpu), 1 GPU path, 2 finished on the host
// after the GPU path failed twice
extern "C" unsigned long long bee2hip_internal_stat(int which)
{
    if (which == 4) return bee2hip::g_new_calls.load();               // operator new calls of the library so far (tune 24's clock)
    if (which == 3) return bee2hip::bign_onekey_table_builds();       // key tables built so far (one-signer / few-signers verification)
    return which == 0 ? bee2hip::g_n_host.load() : which == 1 ? bee2hip::g_n_gpu.load() : bee2hip::g_n_fallback.load();
}

// shader-clock probe: one wavefront spins for `us` microseconds of s_memrealtime (100 MHz) and reports how many
// shader cycles (s_memtime) went by -- launched on a second stream beside the kernels under test, it gives the
// clock the chip actually sustained under that load (DVFS: MI355X_MICROARCH.md "DVFS give-back")
__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(32);
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    } while (r1 - r0 < ticks);
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
extern "C" err_t bee2hip_internal_clock_probe(void *d_out16, unsigned us, void *stream)
try {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), (unsigned long long *)d_out16,
                       (unsigned long long)us * 100ull);
    B2H_TRY(hipGetLastError());
    return ERR_OK;
} B2H_CATCH

// ============================================================ kernel timing ===
extern "C" err_t bee2hip_time_kernel(int which, int reps, void *d_a, void *d_b, void *d_c, void *d_d,
                                     size_t n, size_t aux, void *stream, float *ms)
try {
    if (reps <= 0 || !ms) return ERR_BAD_INPUT;
    err_t code = ensure_device();
    if (code != ERR_OK) return code;
    hipStream_t st = as_stream(stream);
    const octet *H = host_beltH();
    u32 kw[8], c0[4];
    beltKeyExpand2(kw, H + 128, 32);
    for (int i = 0; i < 4; ++i) c0[i] = load32le(H + 192 + 4 * i);
    struct Events {                       // destroyed on every return path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    B2H_TRY(hipEventCreate(&ev.e0));
    B2H_TRY(hipEventCreate(&ev.e1));
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    B2H_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps && code == ERR_OK; ++r) {
        switch (which) {
        case 0: code = launch_bashF_batch(d_a, n, st); break;
        case 1: code = launch_belt_ctr_blocks(d_a, n, kw, c0, 0, nullptr, st); break;
        case 2: code = launch_bign_verify(128, k_oid_belt_hash, sizeof k_oid_belt_hash, d_a, d_b, d_c, n, d_d, st); break;
        case 3: code = launch_bashHash_beltMAC(d_a, aux, n, 256, kw, d_b != nullptr, d_c != nullptr, d_b, d_c, st); break;
        default: code = ERR_BAD_INPUT;
        }
    }
    B2H_TRY(hipEventRecord(e1, st));
    B2H_TRY(hipEventSynchronize(e1));
    float total = 0;
    B2H_TRY(hipEventElapsedTime(&total, e0, e1));
    *ms = total / (float)reps;
    return code;
} B2H_CATCH
// ---- fault injection into the library's own allocations (tune 24).  Replacement allocation functions of THIS shared object:
// -Bsymbolic binds the calls of both translation units to them, nothing outside the library allocates through them; they are
// malloc / free underneath, so memory that crosses into libstdc++ (a std::thread's state) is released correctly there.
static void *b2h_new(size_t n, size_t align)
{
    bee2hip::g_new_calls.fetch_add(1, std::memory_order_relaxed);
    if (bee2hip::g_new_fail_in.load(std::memory_order_relaxed) > 0 && bee2hip::g_new_fail_in.fetch_sub(1) == 1) throw std::bad_alloc();
    void *p = nullptr;
    if (align > alignof(std::max_align_t)) { if (posix_memalign(&p, align, n ? n : 1) != 0) p = nullptr; }
    else p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(size_t n) { return b2h_new(n, 0); }
void *operator new[](size_t n) { return b2h_new(n, 0); }
void *operator new(size_t n, std::align_val_t a) { return b2h_new(n, (size_t)a); }
void *operator new[](size_t n, std::align_val_t a) { return b2h_new(n, (size_t)a); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { try { return b2h_new(n, 0); } catch (...) { return nullptr; } }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { try { return b2h_new(n, 0); } catch (...) { return nullptr; } }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, std::align_val_t) noexcept { free(p); }
void operator delete[](void *p, std::align_val_t) noexcept { free(p); }
void operator delete(void *p, size_t, std::align_val_t) noexcept { free(p); }
void operator delete[](void *p, size_t, std::align_val_t) noexcept { free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }
#endif   // BEE2HIP_EXPERIMENTS
