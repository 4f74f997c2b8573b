"""cpu_baseline: the reference itself (oracle/_ref, bee2 compiled by oracle/Makefile) or the oracle port, timed on this box's
host cores on a bounded sample -- rank 0 at N = 1 only.  The only part of the bench that touches oracle/ (as the CPU baseline,
never as the thing measured)."""
import ctypes
import os
import time

import numpy as np

ALL_THREADS_MIN_S = 2.5          # seconds of timed all-threads work per primitive (four primitives + probes: ~20 s of CPU legs in a default run)

def host_cpus():
    """How many host threads this process may really use: os.cpu_count() is the machine, the affinity mask and the cgroup CPU
    quota are this process's share of it.  threads = min of the three; all of them are reported."""
    count = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = count
    quota = None
    try:                                                   # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        try:                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    threads = max(1, min(count, aff, int(quota) if quota and quota >= 1 else count))
    phys = None
    try:                                                   # physical cores behind the logical ones (SMT siblings share an ALU)
        pairs, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip() and cur:
                pairs.add((cur.get("physical id"), cur.get("core id")))
                cur = {}
        phys = len(pairs) or None
    except Exception:
        pass
    return {"cpu_count": count, "affinity": aff, "cgroup_quota_cpus": quota, "threads": threads, "physical_cores": phys}


def spin_scaling(orc, threads):
    """measured: total rate of `threads` threads each running a dependent 64-bit multiply-add chain / the rate of one thread
    (oracle/orc_threads.c orc_spin_rate): how many cores' worth of cycles the box really gives this process"""
    orc.lib.orc_spin_rate.restype = ctypes.c_double
    orc.lib.orc_spin_rate(int(threads), ctypes.c_double(0.5))    # untimed: the pool's threads are created here, and freshly created
    #                                                              threads take a few hundred ms to spread over the CPUs
    r1 = orc.lib.orc_spin_rate(1, ctypes.c_double(0.4))
    # best of three: the first pass after a single-thread phase can run with the woken threads still queued on one CPU
    rt = max(orc.lib.orc_spin_rate(int(threads), ctypes.c_double(0.5)) for _ in range(3))
    return rt / r1 if r1 else None


def cpu_baseline(which, hc):
    """Time the reference (or the oracle port) on one host thread and on hc["threads"] threads of the oracle's persistent pool,
    every thread >= ~100 ms of work per timed pass (its slice repeated: orc_set_slice_reps).  Returns dict."""
    import orclib
    import refgen
    orc = orclib.load()
    cores = hc["threads"]
    cpuflags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    have_avx512 = " avx512f" in cpuflags
    ref = None
    kind = "port"
    variant = "oracle scalar C"
    if refgen.have_ref():
        path = refgen.REF_AVX512_SO if (which == "bashF" and have_avx512 and os.path.exists(refgen.REF_AVX512_SO)) else refgen.REF_SO
        ref = ctypes.CDLL(path)
        kind = "reference"
        variant = "bee2 BASH_AVX512" if path == refgen.REF_AVX512_SO else "bee2 BASH_64 / scalar C"
    fnptr = lambda name: ctypes.cast(getattr(ref, name), ctypes.c_void_p)  # noqa: E731
    H = orc.beltH()
    if "spin" not in hc:
        hc["spin"] = spin_scaling(orc, cores)
    out = {"cores": cores, "kind": kind, "impl": variant, "cpu_count": hc["cpu_count"], "affinity": hc["affinity"],
           "cgroup_quota_cpus": hc["cgroup_quota_cpus"], "physical_cores": hc["physical_cores"], "spin_scaling": hc["spin"]}

    def clock(run, units, min_s=2.0, max_reps=64):
        reps, t0 = 0, time.perf_counter()
        while True:
            run()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= min_s or reps >= max_reps:
                return units * reps / dt, reps

    def both(run, n, min_s=ALL_THREADS_MIN_S):
        """run(threads) over n units -> (all-threads rate, single-thread rate, note).  A short untimed single-thread probe sizes
        the slice repetitions so that a pass is >= ~100 ms per thread in both legs."""
        orc.lib.orc_set_slice_reps(1)
        run(1)                                             # warm caches / lazy init (the reference's curve object)
        t0 = time.perf_counter()
        run(1)
        t_unit = (time.perf_counter() - t0) / n            # seconds per unit on one thread
        reps1 = max(1, int(0.1 / max(t_unit * n, 1e-9)) + 1)
        repsN = max(1, int(0.1 / max(t_unit * n / cores, 1e-9)) + 1)
        try:
            orc.lib.orc_set_slice_reps(reps1)
            v1, _ = clock(lambda: run(1), n * reps1, min_s=1.0, max_reps=8)
            orc.lib.orc_set_slice_reps(repsN)
            run(cores)                                     # the pool's threads exist from here on
            vall, passes = clock(lambda: run(cores), n * repsN, min_s=min_s)
        finally:
            orc.lib.orc_set_slice_reps(1)
        scal = vall / v1
        note = None
        if scal < 0.7 * cores:
            note = (f"{cores} threads give {scal:.1f}x one thread; a dependent-multiply spin loop on the same pool gives "
                    f"{hc['spin']:.1f}x: that is what the box lets this process have (SMT siblings / shared vCPUs / clocks), "
                    "not a property of the code")
        return vall, v1, scal, f"{passes} timed passes, each thread its slice x {repsN} (>= 100 ms per thread per pass), persistent pool", note

    if which == "bashF":
        n = 1 << 20
        buf = np.empty(192 * n, dtype=np.uint8)
        orc.fill_np(buf, 0xBA5F)
        p = ctypes.c_void_p(buf.ctypes.data)
        if ref is not None:
            run = lambda th: orc.lib.orc_drive_ref_bashF(fnptr("bashF"), p, ctypes.c_size_t(n), th)  # noqa: E731
        else:
            run = lambda th: orc.lib.orc_bashF_batch(p, ctypes.c_size_t(n), th)  # noqa: E731
        vall, v1, scal, how, note = both(run, n)
        out.update(value=vall, unit="perms/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"the 2^20-state batch, {cores} threads over disjoint slices; {how}", scaling_note=note)
        if ref is not None and variant != "bee2 BASH_64 / scalar C":
            ref64 = ctypes.CDLL(refgen.REF_SO)
            f64 = ctypes.cast(ref64.bashF, ctypes.c_void_p)
            v64, v64_1, s64, _, _ = both(lambda th: orc.lib.orc_drive_ref_bashF(f64, p, ctypes.c_size_t(n), th), n, min_s=1.5)
            out["bash64_all_cores"] = v64
            out["bash64_single_thread"] = v64_1
            out["bash64_scaling_over_single_thread"] = s64
    elif which == "ctr":
        nbytes = 256 << 20
        buf = np.zeros(nbytes, dtype=np.uint8)
        kw, c0 = orc.ctr_start(H[128:160], H[192:208])
        p = ctypes.c_void_p(buf.ctypes.data)
        nb = nbytes // 16
        if ref is not None:
            run = lambda th: orc.lib.orc_drive_ref_ctr(fnptr("beltCTRStepE"), p, ctypes.c_size_t(nb), kw, c0, ctypes.c_uint64(0), th)  # noqa: E731
        else:
            run = lambda th: orc.lib.orc_beltCTR_blocks(p, ctypes.c_size_t(nb), kw, c0, ctypes.c_uint64(0), th)  # noqa: E731
        # (the single-thread leg runs over a 16 MiB prefix: 256 MiB would be 1.3 s per pass)
        nb1 = nb // 16
        run1 = lambda: (orc.lib.orc_drive_ref_ctr(fnptr("beltCTRStepE"), p, ctypes.c_size_t(nb1), kw, c0, ctypes.c_uint64(0), 1)  # noqa: E731
                        if ref is not None else orc.lib.orc_beltCTR_blocks(p, ctypes.c_size_t(nb1), kw, c0, ctypes.c_uint64(0), 1))
        orc.lib.orc_set_slice_reps(1)
        run1()
        v1, _ = clock(run1, nb1 * 16 / 2 ** 30, min_s=1.0, max_reps=16)
        t_unit = 1.0 / (v1 * 2 ** 30 / 16)                 # seconds per block, one thread
        repsN = max(1, int(0.1 / (t_unit * nb / cores)) + 1)
        try:
            orc.lib.orc_set_slice_reps(repsN)
            run(cores)
            vall, passes = clock(lambda: run(cores), nbytes * repsN / 2 ** 30, min_s=ALL_THREADS_MIN_S)
        finally:
            orc.lib.orc_set_slice_reps(1)
        scal = vall / v1
        out.update(value=vall, unit="GiB/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"{passes} timed passes over a 256 MiB prefix of the stream, {cores} threads, each its slice x {repsN}; one thread: a 16 MiB prefix",
                   scaling_note=None if scal >= 0.7 * cores else f"{scal:.1f}x over one thread against {hc['spin']:.1f}x for a spin loop on the same pool")
    elif which == "verify":
        G = orclib.Golden()
        hs, ss, ps = G.bign_base_arrays()
        nbase = len(hs) // 32
        tile = max(1, min(64, (cores * 64 + nbase - 1) // nbase))   # >= 64 signatures per thread: 2048 x tile
        hs, ss, ps = hs * tile, ss * tile, ps * tile
        n = nbase * tile
        codes = (ctypes.c_uint32 * n)()
        if ref is not None:
            ref.bign128Verify.restype = ctypes.c_uint32
            run = lambda th, cnt: orc.lib.orc_drive_ref_verify(fnptr("bign128Verify"), hs, ss, ps, ctypes.c_size_t(cnt), codes, th)  # noqa: E731
        else:
            run = lambda th, cnt: orc.lib.orc_bign128Verify_batch(hs, ss, ps, ctypes.c_size_t(cnt), codes, th)  # noqa: E731
        # the single-thread probe runs over the first 256 signatures, the all-threads leg over all n: the item count is explicit in
        # both calls (a box that gives this process ONE thread runs both legs with th == 1 and still counts what it processed)
        orc.lib.orc_set_slice_reps(1)
        run(1, 256)
        v1, _ = clock(lambda: run(1, 256), 256, min_s=1.0, max_reps=16)
        repsN = max(1, int(0.1 / (n / cores / v1)) + 1)
        try:
            orc.lib.orc_set_slice_reps(repsN)
            run(cores, n)
            vall, passes = clock(lambda: run(cores, n), n * repsN, min_s=ALL_THREADS_MIN_S)
        finally:
            orc.lib.orc_set_slice_reps(1)
        assert all(c == 0 for c in codes)
        scal = vall / v1
        out.update(value=vall, unit="verifies/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"{passes} timed passes over the 2048 genuine signatures of tests/golden/bign_base.bin tiled x {tile}, {cores} threads, "
                          f"each its slice x {repsN} (>= 100 ms per thread per pass); one thread: the first 256",
                   scaling_note=None if scal >= 0.7 * cores else f"{scal:.1f}x over one thread against {hc['spin']:.1f}x for a spin loop on the same pool")
    elif which == "mixed":
        ml = 4096
        n = max(1 << 12, min(1 << 16, cores * 64))         # >= 64 messages per thread
        msgs = orc.fill(n * ml, 0x4D1C)
        dig = ctypes.create_string_buffer(64 * n)
        tag = ctypes.create_string_buffer(8 * n)
        key = H[128:160]
        if ref is not None:
            run = lambda th, cnt: orc.lib.orc_drive_ref_mixed(fnptr("bashHash"), fnptr("beltMAC"), msgs, ctypes.c_size_t(ml), ctypes.c_size_t(cnt), key, ctypes.c_size_t(32), dig, tag, th)  # noqa: E731
        else:
            run = lambda th, cnt: orc.lib.orc_bash512_beltMAC_batch(msgs, ctypes.c_size_t(ml), ctypes.c_size_t(cnt), key, ctypes.c_size_t(32), dig, tag, th)  # noqa: E731
        orc.lib.orc_set_slice_reps(1)
        run(1, 256)
        v1, _ = clock(lambda: run(1, 256), 256, min_s=1.0, max_reps=64)
        repsN = max(1, int(0.1 / (n / cores / v1)) + 1)
        try:
            orc.lib.orc_set_slice_reps(repsN)
            run(cores, n)
            vall, passes = clock(lambda: run(cores, n), n * repsN, min_s=ALL_THREADS_MIN_S)
        finally:
            orc.lib.orc_set_slice_reps(1)
        scal = vall / v1
        out.update(value=vall, unit="messages/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"{passes} timed passes over {n} x 4 KiB messages, {cores} threads, each its slice x {repsN}; one thread: the first 256",
                   scaling_note=None if scal >= 0.7 * cores else f"{scal:.1f}x over one thread against {hc['spin']:.1f}x for a spin loop on the same pool")
    return out

