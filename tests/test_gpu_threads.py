"""-m gpu: the drop-in entry points are re-entrant, as bee2's are (SURVEY.md 8b "Threading"): several host
threads call a mix of them at the same time (ctypes releases the GIL for the duration of a call, so the
calls really overlap inside the library: per-thread scratch, the per-stream pool, lazily built tables)."""
import threading

import pytest

from gpulib import engine

pytestmark = pytest.mark.gpu


def test_dropin_calls_from_eight_threads(orc, golden):
    eng = engine()
    H = golden.H
    hs, ss, ps = golden.bign_base_arrays()
    import os
    nthreads, rounds = int(os.environ.get("BEE2_TEST_THREADS", "8")), 6
    jobs = []                                     # per thread: list of (callable, expected)
    for t in range(nthreads):
        key, iv = orc.fill(32, 100 + t), orc.fill(16, 200 + t)
        msg = orc.fill(16 * (40 + 7 * t) + (t % 5), 300 + t)          # ragged for CTR / MAC / hash
        blocks = orc.fill(16 * (64 + t), 400 + t)                      # whole blocks for the block modes
        i = 17 * t
        trip = (hs[32 * i:32 * i + 32], ss[48 * i:48 * i + 48], ps[64 * i:64 * i + 64])
        bad = (trip[0], bytes([trip[1][0] ^ 1]) + trip[1][1:], trip[2])
        crit, op = msg[:100 + t], msg[100 + t:]
        wrapped = orc.dwp_wrap(crit, op, key, iv)
        che = orc.dwp_wrap(crit, op, key, iv, "CHE")
        jobs.append([
            (lambda m=msg, k=key, v=iv: eng.beltCTR(m, k, v), (0, orc.ctr(msg, key, iv))),
            (lambda m=msg: eng.bashHash(128, m), orc.bashHash(128, msg)),
            (lambda m=msg, k=key: eng.beltMAC(m, k), (0, orc.mac(msg, key))),
            (lambda b=blocks, k=key, v=iv: eng.belt_mode("beltBDEEncr", b, k, v), orc.bde(blocks, key, iv)),
            (lambda b=blocks, k=key, v=iv: eng.belt_mode("beltSDEDecr", b, k, v), orc.sde(blocks, key, iv, True)),
            (lambda b=blocks, k=key, v=iv: eng.belt_mode("beltCBCDecr", b, k, v), orc.cbc(blocks, key, iv, True)),
            (lambda c=crit, o=op, k=key, v=iv: eng.dwp_wrap(c, o, k, v), wrapped),
            (lambda c=che[1], o=op, m=che[2], k=key, v=iv: eng.dwp_unwrap(c, o, m, k, v, "CHE"), (0, crit)),
            (lambda x=trip: eng.bign128Verify(*x), 0),
            (lambda x=bad: eng.bign128Verify(*x), 510),
            (lambda m=msg: eng.hash_ragged(0, [m, m[:33], b""]),
             (0, [orc.belt_hash(msg), orc.belt_hash(msg[:33]), orc.belt_hash(b"")])),
        ])
    errors = []
    start = threading.Barrier(nthreads)

    def run(t):
        try:
            start.wait()
            for r in range(rounds):
                order = jobs[t][r % len(jobs[t]):] + jobs[t][:r % len(jobs[t])]     # threads are out of phase
                for j, (call, want) in enumerate(order):
                    got = call()
                    if got != want:
                        errors.append((t, r, j))
                        return
        except Exception as e:                                            # noqa: BLE001
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a thread is stuck"
    assert not errors, errors[:5]


def test_short_lived_threads_do_not_leak_device_memory(orc, golden):
    """a thread-per-request caller of the drop-in API: per-thread staging and NULL-stream scratch are released
    when the thread exits (they used to stay allocated for ever)"""
    import threading

    import torch

    from gpulib import engine
    eng = engine()
    h, s, p = golden.bign_base[0]
    key = golden.H[128:160]

    def one():
        eng.set_device(0)
        assert eng.bign128Verify(h, s, p) == 0
        eng.beltMAC(bytes(1000), key)
        eng.beltHash(bytes(70000))

    def burst(k):
        for _ in range(k):
            t = threading.Thread(target=one)
            t.start()
            t.join()
    burst(8)                                   # tables, first-use allocations
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    burst(150)
    torch.cuda.synchronize()
    import time
    for _ in range(50):                        # (the last thread's thread_local destructors run just after join() returns)
        free1, _ = torch.cuda.mem_get_info()
        if free0 - free1 < (4 << 20):
            break
        time.sleep(0.1)
    assert free0 - free1 < (4 << 20), f"device memory shrank by {(free0 - free1) >> 10} KiB over 150 threads"


def test_threads_that_ran_the_chunked_host_pipelines_give_their_scratch_back(orc, golden):
    """a host-pointer verification of 2^19 signatures and more, and a CTR call of 48 MiB and more, run on
    library-owned per-thread streams; the launchers' scratch keyed on those streams (275 MB for a 2^18-signature chunk) must be
    released with the streams when the thread exits -- it used to stay for the rest of the process."""
    import threading

    import numpy as np
    import torch

    from gpulib import engine
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    nb = len(hs) // 32
    reps = (1 << 19) // nb
    H, S, K = hs * reps, ss * reps, ps * reps
    key, iv = golden.H[128:160], golden.H[192:208]
    big = np.zeros(64 << 20, dtype=np.uint8)
    bad = []

    def one():
        eng.set_device(0)
        code, codes = eng.bignVerify_batch(H, S, K)
        if code != 0 or any(codes):
            bad.append(code)
        if eng.lib.beltCTR(big.ctypes.data_as(__import__("ctypes").c_void_p), big.ctypes.data_as(__import__("ctypes").c_void_p),
                           __import__("ctypes").c_size_t(big.nbytes), key, __import__("ctypes").c_size_t(32), iv) != 0:
            bad.append("ctr")

    def burst(k):
        for _ in range(k):
            t = threading.Thread(target=one)
            t.start()
            t.join()
    burst(2)                                   # tables, first-use allocations
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    burst(6)
    torch.cuda.synchronize()
    # (Thread.join returns when the Python side of the thread is through; the C++ thread_local destructors that give the blocks back run
    #  as the OS thread exits, a moment later: wait for the last one instead of racing it)
    import time
    for _ in range(50):
        free1, _ = torch.cuda.mem_get_info()
        if free0 - free1 < (32 << 20):
            break
        time.sleep(0.1)
    assert not bad, bad
    assert free0 - free1 < (32 << 20), f"device memory shrank by {(free0 - free1) >> 20} MiB over 6 threads"


def test_ragged_batches_on_two_queues_from_six_threads(orc):
    """bee2hip_hash_ragged with 1200-1700 messages per call (long chains on the thread's stream, short messages on ITS side
    stream -- thread-local, created on first use, released when the thread exits) from six threads at once, three rounds each,
    belt-hash and bash256 alternating: every digest against the oracle; then the threads are gone and six NEW ones do it again
    (their side streams are new objects: nothing of the old ones may be reused)."""
    import random
    eng = engine()
    batches = []
    for t in range(6):
        rnd = random.Random(900 + t)
        msgs = [rnd.randbytes(rnd.choice((0, 3, 32, 33, 200, 999))) for _ in range(1200 + 100 * t)]
        for k in rnd.sample(range(len(msgs)), 5):
            msgs[k] = rnd.randbytes(rnd.choice((4096, 9000, 40001)))
        want = {0: [orc.belt_hash(m) for m in msgs], 128: [orc.bashHash(128, m)[1] for m in msgs]}
        batches.append((msgs, want))
    for wave in range(2):
        errors = []
        start = threading.Barrier(6)

        def run(t):
            try:
                msgs, want = batches[t]
                start.wait()
                for r in range(3):
                    alg = (0, 128)[(r + t) & 1]
                    code, got = eng.hash_ragged(alg, msgs)
                    if code != 0 or got != want[alg]:
                        errors.append((wave, t, r, code))
                        return
            except Exception as e:                                            # noqa: BLE001
                errors.append((wave, t, repr(e)))
        threads = [threading.Thread(target=run, args=(t,)) for t in range(6)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not any(th.is_alive() for th in threads), "a thread is stuck"
        assert not errors, errors[:5]
