"""-m gpu: the *_multi host entry points (one process, one worker thread per device).  On a one-GPU
box BEE2HIP_FAKE_DEVICES makes the library run k logical devices on the one real GPU, so the partition, the worker
threads and the CTR state hand-over are exercised; on a multi-GPU box the same tests use the real devices too."""
import ctypes
import os

import numpy as np
import pytest

from bee2_amd import engine as E
from gpulib import engine

pytestmark = pytest.mark.gpu
_sz = ctypes.c_size_t


@pytest.fixture(params=[1, 3, 8])
def devices(request):
    os.environ["BEE2HIP_FAKE_DEVICES"] = str(request.param)
    yield request.param
    os.environ.pop("BEE2HIP_FAKE_DEVICES", None)


def test_device_count_and_fake(devices):
    eng = engine()
    assert eng.lib.bee2hip_device_count() == devices


def test_bashF_and_mixed_multi(orc, golden, devices):
    eng = engine()
    n = 10_001
    data = orc.fill(192 * n, 0xBA5F)
    buf = ctypes.create_string_buffer(data, len(data))
    assert eng.lib.bee2hip_bashF_batch_multi(buf, _sz(n), 0) == 0
    assert buf.raw == orc.bashF_batch(data, nthreads=8)
    m, ml = 2_003, 1024
    msgs = orc.fill(m * ml, 0x4D1C)
    key = golden.H[128:160]
    dig = ctypes.create_string_buffer(64 * m)
    tag = ctypes.create_string_buffer(8 * m)
    assert eng.lib.bee2hip_bashHash_beltMAC_batch_multi(msgs, _sz(ml), _sz(m), _sz(256), key, _sz(32), dig, tag, 0) == 0
    wd, wt = orc.mixed_batch(msgs, ml, key, nthreads=8)
    assert dig.raw == wd and tag.raw == wt


def test_ctr_multi_leaves_the_state_of_the_serial_call(orc, golden, devices):
    eng = engine()
    key, iv = golden.H[128:160], golden.H[192:208]
    msg = orc.fill(1_000_003, 0xBE17)                 # ragged tail: 3 bytes into the last block
    want = orc.ctr(msg, key, iv)
    for cut in (0, 5, 4096 + 7):                       # leading piece through the single-device entry, left-over gamma
        st1 = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
        st2 = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
        eng.lib.beltCTRStart(st1, key, _sz(32), iv)
        eng.lib.beltCTRStart(st2, key, _sz(32), iv)
        a = ctypes.create_string_buffer(msg, len(msg))
        b = ctypes.create_string_buffer(msg, len(msg))
        if cut:
            assert eng.lib.bee2hip_beltCTR_bulk(a, _sz(cut), st1) == 0
            assert eng.lib.bee2hip_beltCTR_bulk(b, _sz(cut), st2) == 0
        assert eng.lib.bee2hip_beltCTR_bulk(ctypes.byref(a, cut), _sz(len(msg) - cut), st1) == 0
        assert eng.lib.bee2hip_beltCTR_bulk_multi(ctypes.byref(b, cut), _sz(len(msg) - cut), st2, 0) == 0
        assert a.raw == want and b.raw == want
        assert st1.raw == st2.raw                      # counter, gamma block, reserved: identical
        # and the stream continues correctly from that state
        more = ctypes.create_string_buffer(bytes(100), 100)
        more2 = ctypes.create_string_buffer(bytes(100), 100)
        eng.lib.beltCTRStepE(more, _sz(100), st1)
        eng.lib.beltCTRStepE(more2, _sz(100), st2)
        assert more.raw == more2.raw


def test_verify_sign_and_ragged_multi(orc, golden, devices):
    eng = engine()
    P = eng.bignParamsStd(E.CURVE_NAME[128])
    oid = E.LEVEL_OID[128]
    hs, ss, ps = golden.bign_base_arrays()
    n = 1500
    bad = bytearray(ss[:48 * n])
    for i in range(0, n, 7):
        bad[48 * i + 3] ^= 1
    codes = (ctypes.c_uint32 * n)()
    assert eng.lib.bee2hip_bignVerify_batch_multi(ctypes.byref(P), oid, _sz(11), hs[:32 * n], bytes(bad), ps[:64 * n], _sz(n), codes, 0) == 0
    assert list(codes) == orc.verify_batch(hs[:32 * n], bytes(bad), ps[:64 * n], nthreads=8)
    privs = orc.fill(32 * 300, 0x51)
    sig = ctypes.create_string_buffer(48 * 300)
    sc = (ctypes.c_uint32 * 300)()
    assert eng.lib.bee2hip_bignSign2_batch_multi(ctypes.byref(P), oid, _sz(11), hs[:32 * 300], privs, None, _sz(0), _sz(300), sig, sc, 0) == 0
    for i in (0, 1, 99, 100, 101, 299):
        assert orc.sign2(128, oid, hs[32 * i: 32 * i + 32], privs[32 * i: 32 * i + 32], None) == (sc[i], sig.raw[48 * i: 48 * i + 48])
    # argument errors surface as from the single-device call
    assert eng.lib.bee2hip_bignVerify_batch_multi(ctypes.byref(P), b"\x06\x01", _sz(2), hs, ss, ps, _sz(4), codes, 0) == E.ERR_BAD_OID
    # ragged: byte-balanced ranges of whole messages
    import random
    rnd = random.Random(devices)
    msgs = [rnd.randbytes(rnd.choice((0, 1, 31, 32, 33, 500, 5000, 70_000))) for _ in range(400)]
    data = b"".join(msgs)
    offs = np.zeros(len(msgs) + 1, dtype=np.uint64)
    np.cumsum([len(m) for m in msgs], out=offs[1:])
    for alg, dlen in ((0, 32), (128, 32), (256, 64)):
        out = ctypes.create_string_buffer(dlen * len(msgs))
        assert eng.lib.bee2hip_hash_ragged_multi(_sz(alg), data, offs.ctypes.data_as(ctypes.c_void_p), _sz(len(msgs)), out, 0) == 0
        for i, m in enumerate(msgs):
            want = orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1]
            assert out.raw[dlen * i: dlen * (i + 1)] == want, (alg, i)


def test_multi_dev_entries_on_resident_shards(orc, golden, devices):
    """shards already in device memory, one pointer + count per device, no staging through the host.
    With BEE2HIP_FAKE_DEVICES the k logical devices are the one real GPU, so every shard is a tensor on cuda:0; the CTR
    shards are consecutive pieces of ONE stream (first_block offsets computed by the library)."""
    import torch

    from gpulib import dev, host
    eng = engine()
    k = devices
    vp = ctypes.c_void_p
    # uneven shards, one of them empty when there are enough devices
    counts = [(1000 + 137 * i) if (i != 1 or k < 3) else 0 for i in range(k)]
    CNT = (ctypes.c_size_t * k)(*counts)
    # bashF
    data = [orc.fill(192 * c, 0xBA50 + i) for i, c in enumerate(counts)]
    ts = [dev(d) if d else torch.empty(0, dtype=torch.uint8, device="cuda") for d in data]
    P = (vp * k)(*[t.data_ptr() if t.numel() else None for t in ts])
    assert eng.lib.bee2hip_bashF_batch_multi_dev(P, CNT, k) == 0
    for t, d in zip(ts, data):
        assert host(t) == orc.bashF_batch(d)
    # CTR: one stream of sum(counts) blocks, sharded; starts 5 blocks into the stream
    H = golden.H
    kw, c0 = eng.beltCTRStart(H[128:160], H[192:208])
    total = sum(counts)
    stream = np.frombuffer(orc.fill(16 * total, 0xBE17), dtype=np.uint8).copy()
    want = stream.copy()
    orc.ctr_blocks_np(want, kw, c0, first=5)
    ts, pos = [], 0
    for c in counts:
        ts.append(dev(stream[16 * pos: 16 * (pos + c)]) if c else torch.empty(0, dtype=torch.uint8, device="cuda"))
        pos += c
    P = (vp * k)(*[t.data_ptr() if t.numel() else None for t in ts])
    assert eng.lib.bee2hip_beltCTR_blocks_multi_dev(P, CNT, bytes(kw), bytes(c0), ctypes.c_uint64(5), k) == 0
    assert b"".join(host(t) for t in ts) == want.tobytes()
    # verify: the base set cut into shards, every 5th signature corrupted
    hs, ss, ps = golden.bign_base_arrays()
    vc = [min(c, 250) for c in counts]              # 8 x 250 <= the 2048 triples of the base set
    VC = (ctypes.c_size_t * k)(*vc)
    bad = bytearray(ss)
    for i in range(0, len(ss) // 48, 5):
        bad[48 * i + 7] ^= 4
    th, tsg, tp, tc, pos = [], [], [], [], 0
    for c in vc:
        th.append(dev(hs[32 * pos: 32 * (pos + c)]) if c else None)
        tsg.append(dev(bytes(bad[48 * pos: 48 * (pos + c)])) if c else None)
        tp.append(dev(ps[64 * pos: 64 * (pos + c)]) if c else None)
        tc.append(torch.full((max(c, 1),), -1, dtype=torch.int32, device="cuda"))
        pos += c
    arr = lambda xs: (vp * k)(*[x.data_ptr() if x is not None else None for x in xs])  # noqa: E731
    oid = E.LEVEL_OID[128]
    assert eng.lib.bee2hip_bignVerifyL_batch_multi_dev(_sz(128), oid, _sz(11), arr(th), arr(tsg), arr(tp), VC, arr(tc), k) == 0
    got = [int(x) & 0xFFFFFFFF for t, c in zip(tc, vc) for x in t.cpu().numpy()[:c]]
    assert got == orc.verify_batch(hs[:32 * pos], bytes(bad[:48 * pos]), ps[:64 * pos], nthreads=8)
    # bash512 + beltMAC per message
    ml = 512
    mc = [min(c, 300) for c in counts]
    MC = (ctypes.c_size_t * k)(*mc)
    msgs = [orc.fill(ml * c, 0x4D10 + i) for i, c in enumerate(mc)]
    tm = [dev(m) if m else None for m in msgs]
    td = [torch.zeros(64 * max(c, 1), dtype=torch.uint8, device="cuda") for c in mc]
    tt = [torch.zeros(8 * max(c, 1), dtype=torch.uint8, device="cuda") for c in mc]
    key = H[128:160]
    assert eng.lib.bee2hip_bashHash_beltMAC_batch_multi_dev(arr(tm), _sz(ml), MC, _sz(256), key, _sz(32), arr(td), arr(tt), k) == 0
    for m, d, t, c in zip(msgs, td, tt, mc):
        if c:
            wd, wt = orc.mixed_batch(m, ml, key)
            assert host(d)[: 64 * c] == wd and host(t)[: 8 * c] == wt
    # misuse: more entries than devices, null arrays
    assert eng.lib.bee2hip_bashF_batch_multi_dev(P, CNT, k + 1) == E.ERR_BAD_INPUT
    assert eng.lib.bee2hip_bashF_batch_multi_dev(None, CNT, k) == E.ERR_BAD_INPUT


def test_onekey_and_keyed_multi(orc, golden, devices):
    """the one-signer / few-signers entries over k (logical) devices: host-pointer forms cut by bee2hip_multi_plan, device-resident
    shards with the keys on the host -- the base set's signatures by their signers, every 6th damaged, the oracle's verdicts"""
    import torch

    from gpulib import dev
    eng = engine()
    k = devices
    vp = ctypes.c_void_p
    P = eng.bignParamsStd(E.CURVE_NAME[128])
    oid = E.LEVEL_OID[128]
    hs, ss, ps = golden.bign_base_arrays()
    n = 1800
    bad = bytearray(ss[:48 * n])
    for i in range(0, n, 6):
        bad[48 * i + 11] ^= 2
    keys = sorted({ps[64 * i: 64 * i + 64] for i in range(n)})
    idx = np.array([keys.index(ps[64 * i: 64 * i + 64]) for i in range(n)], dtype=np.uint32)
    want = orc.verify_batch(hs[:32 * n], bytes(bad), ps[:64 * n], nthreads=8)
    codes = (ctypes.c_uint32 * n)()
    assert eng.lib.bee2hip_bignVerify_keyed_batch_multi(ctypes.byref(P), oid, _sz(11), hs[:32 * n], bytes(bad), b"".join(keys), _sz(len(keys)),
                                                        idx.ctypes.data_as(vp), _sz(n), codes, 0) == 0
    assert list(codes) == want
    # one signer: the signatures of the first key, and as the wrong key for the others
    mine = [i for i in range(n) if idx[i] == idx[0]]
    H1 = b"".join(hs[32 * i: 32 * i + 32] for i in mine) * 9
    S1 = b"".join(bytes(bad[48 * i: 48 * i + 48]) for i in mine) * 9
    m = len(mine) * 9
    c1 = (ctypes.c_uint32 * m)()
    assert eng.lib.bee2hip_bignVerify_onekey_batch_multi(ctypes.byref(P), oid, _sz(11), H1, S1, keys[idx[0]], _sz(m), c1, 0) == 0
    assert list(c1) == [want[i] for i in mine] * 9
    assert eng.lib.bee2hip_bignVerify_onekey_batch_multi(ctypes.byref(P), b"\x06\x01", _sz(2), H1, S1, keys[idx[0]], _sz(m), c1, 0) == E.ERR_BAD_OID
    # resident shards
    counts = [(100 + 17 * i) if (i != 1 or k < 3) else 0 for i in range(k)]
    CNT = (ctypes.c_size_t * k)(*counts)
    th, tsg, ti, tc, pos = [], [], [], [], 0
    for c in counts:
        th.append(dev(hs[32 * pos: 32 * (pos + c)]) if c else None)
        tsg.append(dev(bytes(bad[48 * pos: 48 * (pos + c)])) if c else None)
        ti.append(torch.from_numpy(idx[pos: pos + c].astype(np.int32)).cuda() if c else None)
        tc.append(torch.full((max(c, 1),), -1, dtype=torch.int32, device="cuda"))
        pos += c
    arr = lambda xs: (vp * k)(*[x.data_ptr() if x is not None else None for x in xs])  # noqa: E731
    assert eng.lib.bee2hip_bignVerifyL_keyed_batch_multi_dev(_sz(128), oid, _sz(11), arr(th), arr(tsg), b"".join(keys), _sz(len(keys)), arr(ti),
                                                             CNT, arr(tc), k) == 0
    got = [int(x) & 0xFFFFFFFF for t, c in zip(tc, counts) for x in t.cpu().numpy()[:c]]
    assert got == want[:pos]
    for t in tc:
        t.fill_(-1)
    assert eng.lib.bee2hip_bignVerifyL_onekey_batch_multi_dev(_sz(128), oid, _sz(11), arr(th), arr(tsg), keys[idx[0]], CNT, arr(tc), k) == 0
    got = [int(x) & 0xFFFFFFFF for t, c in zip(tc, counts) for x in t.cpu().numpy()[:c]]
    assert got == [want[i] if idx[i] == idx[0] else 510 for i in range(pos)]


def test_logical_devices_that_share_a_card_run_side_by_side(orc, golden):
    """the N > 1 library path cannot meet a second card here, so it meets a second QUEUE.  Every pool worker
    launches its _multi_dev job on its own non-blocking stream: four latency-bound shards (2^9 signatures each: 64 wavefronts, a
    chain of 0.36 ms whatever the size, on a chip with 1024 SIMDs) on four logical devices of ONE card must overlap, not take turns as
    they did on the card's NULL stream (4 x the time of one shard).  Measured (tools/ab/multi_ratio_probe.py): 0.41 ms for one shard, 0.81
    for four = x2.0 -- the rest is the host side of four workers launching five kernels each; bigger shards (2^12: x2.5) also meet
    in the VALU, which one wavefront of these kernels nearly fills."""
    import time

    import torch

    from gpulib import dev
    os.environ["BEE2HIP_FAKE_DEVICES"] = "4"
    try:
        eng = engine()
        vp = ctypes.c_void_p
        hs, ss, ps = golden.bign_base_arrays()
        m = 1 << 9
        bad = bytearray(ss * 2)
        for i in range(0, m, 7):
            bad[48 * i + 3] ^= 1
        th, tsg, tp = dev((hs * 2)[: 32 * m]), dev(bytes(bad[: 48 * m])), dev((ps * 2)[: 64 * m])
        want = torch.tensor(orc.verify_batch((hs * 2)[: 32 * m], bytes(bad[: 48 * m]), (ps * 2)[: 64 * m], nthreads=8), dtype=torch.int64)
        oid = E.LEVEL_OID[128]

        def run(k):
            tc = [torch.full((m,), -1, dtype=torch.int32, device="cuda") for _ in range(k)]
            arr = lambda xs: (vp * k)(*[x.data_ptr() for x in xs])  # noqa: E731
            CNT = (ctypes.c_size_t * k)(*([m] * k))
            args = (_sz(128), oid, _sz(11), arr([th] * k), arr([tsg] * k), arr([tp] * k), CNT, arr(tc), k)
            torch.cuda.synchronize()
            assert eng.lib.bee2hip_bignVerifyL_batch_multi_dev(*args) == 0            # warm (worker threads, streams, scratch)
            best = 1e9
            for _ in range(7):
                t0 = time.perf_counter()
                assert eng.lib.bee2hip_bignVerifyL_batch_multi_dev(*args) == 0
                best = min(best, time.perf_counter() - t0)
            for t in tc:                                           # the call has drained the workers' streams: results are there
                assert torch.equal(t.cpu().to(torch.int64) & 0xFFFFFFFF, want)
            return best
        t1, t4 = run(1), run(4)
        assert t4 < 3.0 * t1, (t1, t4)                             # taking turns would be ~4x; measured x2.0
    finally:
        os.environ.pop("BEE2HIP_FAKE_DEVICES", None)


def test_two_host_threads_on_two_multi_entries_take_turns_and_both_finish(orc, golden):
    """the pool runs ONE multi-device batch at a time (multi.hip call_mu: its workers are one thread per device): two host
    threads that call two different _multi entries at once are serialised, not interleaved -- both must return the right
    bytes, repeatedly"""
    import threading
    os.environ["BEE2HIP_FAKE_DEVICES"] = "3"
    try:
        eng = engine()
        H = golden.H
        n = 5000
        states = orc.fill(192 * n, 0xBA5F)
        want_f = orc.bashF_batch(states)
        stream = np.frombuffer(orc.fill(16 * 70001, 0xBE17), dtype=np.uint8).copy()
        kw, c0 = orc.ctr_start(H[128:160], H[192:208])
        want_c = stream.copy()
        orc.ctr_blocks_np(want_c, kw, c0, first=0)
        errs = []

        def f_bash():
            try:
                for _ in range(6):
                    buf = ctypes.create_string_buffer(states, len(states))
                    assert eng.lib.bee2hip_bashF_batch_multi(buf, _sz(n), 3) == 0
                    assert buf.raw == want_f
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        def f_ctr():
            try:
                for _ in range(6):
                    st = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
                    eng.lib.beltCTRStart(st, bytes(H[128:160]), _sz(32), bytes(H[192:208]))
                    buf = stream.copy()
                    assert eng.lib.bee2hip_beltCTR_bulk_multi(ctypes.c_void_p(buf.ctypes.data), _sz(buf.size), st, 3) == 0
                    assert np.array_equal(buf, want_c)
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))
        ts = [threading.Thread(target=f_bash), threading.Thread(target=f_ctr)]
        [t.start() for t in ts]
        [t.join(300) for t in ts]
        assert not errs and not any(t.is_alive() for t in ts), errs
    finally:
        os.environ.pop("BEE2HIP_FAKE_DEVICES", None)
