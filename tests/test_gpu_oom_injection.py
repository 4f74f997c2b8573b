"""-m gpu: nothing unwinds through the C ABI.  The experiments build replaces the library's own operator new
(bee2_amd/csrc/capi_exp.hip) and bee2hip_internal_tune(24, n) makes the n-th allocation from now on throw std::bad_alloc: every
entry point walked here must then return an error CODE (ERR_OUTOFMEMORY = 110) -- or succeed, when the call makes fewer than n
allocations -- never crash, never leave a helper thread behind, and leave the caller's buffers as they were on failure."""
import ctypes

import numpy as np
import pytest

from gpulib import exp_engine

pytestmark = pytest.mark.gpu
ERR_OUTOFMEMORY = 110
_sz = ctypes.c_size_t


def _walk(eng, call, untouched, n_max=40, check_ok=None):
    """call() -> code with the n-th allocation failing, n = 1 .. n_max; returns (#failures seen, #allocations of a clean call).
    A call may also SURVIVE a failed allocation (a helper thread that could not be started has its share done inline): then the
    results must be the right ones (check_ok)."""
    eng.lib.bee2hip_internal_stat.restype = ctypes.c_ulonglong
    eng.lib.bee2hip_internal_tune(24, 0)
    assert call() == 0                                        # warm: tables, staging, pools exist
    a0 = eng.lib.bee2hip_internal_stat(4)
    assert call() == 0
    allocs = eng.lib.bee2hip_internal_stat(4) - a0
    failures = 0
    for n in range(1, n_max + 1):
        eng.lib.bee2hip_internal_tune(24, n)
        code = call()
        eng.lib.bee2hip_internal_tune(24, 0)
        assert code in (0, ERR_OUTOFMEMORY), (n, allocs, code)
        if code == ERR_OUTOFMEMORY:
            assert n <= allocs + 2, (n, allocs)
            failures += 1
            untouched()
        else:
            assert n > allocs or check_ok is not None, (n, allocs)
            if check_ok is not None:
                check_ok()
    assert call() == 0                                        # and the library works afterwards
    return failures, allocs


def test_bashF_batch_host_pointer_duplex_pipeline(orc):
    eng = exp_engine()
    n = (48 << 20) // 192 + 1000                              # >= 48 MiB: the duplex pipeline (a vector of cuts, events, a helper thread)
    src = np.frombuffer(orc.fill(192 * n, 0xBA5F), dtype=np.uint8).copy()
    buf = src.copy()

    def call():
        buf[:] = src
        return eng.lib.bee2hip_bashF_batch(ctypes.c_void_p(buf.ctypes.data), _sz(n))

    def untouched():
        assert np.array_equal(buf, src)
    failures, allocs = _walk(eng, call, untouched)
    assert allocs >= 2 and failures == min(allocs, 40)
    want = np.frombuffer(orc.bashF_batch(src[: 192 * 64].tobytes()), dtype=np.uint8)
    assert call() == 0 and np.array_equal(buf[: 192 * 64], want)


def test_one_shot_beltCTR_48MiB(orc, golden):
    eng = exp_engine()
    nbytes = (48 << 20) + 37
    src = np.frombuffer(orc.fill(nbytes + 11, 3), dtype=np.uint8)[:nbytes].copy()
    buf = src.copy()
    key, iv = golden.H[128:160], golden.H[192:208]
    eng.lib.bee2hip_internal_tune(4, 1)                       # as BEE2HIP_FORCE=gpu: an allocation failure is reported, not finished on the host

    def call():
        buf[:] = src
        return eng.lib.beltCTR(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(buf.ctypes.data), _sz(nbytes), bytes(key), _sz(32), bytes(iv))

    def untouched():
        assert np.array_equal(buf, src)
    try:
        failures, allocs = _walk(eng, call, untouched)
    finally:
        eng.lib.bee2hip_internal_tune(4, 0)
    assert allocs >= 2 and failures == min(allocs, 40)
    kw, c0 = orc.ctr_start(key, iv)
    want = src[: 16 * 4096].copy()
    orc.ctr_blocks_np(want, kw, c0, first=0)
    assert call() == 0 and np.array_equal(buf[: 16 * 4096], want)
    # the default mode finishes on the host instead of failing: same bytes, no error
    eng.lib.bee2hip_internal_tune(24, 1)
    assert call() == 0
    eng.lib.bee2hip_internal_tune(24, 0)
    assert np.array_equal(buf[: 16 * 4096], want)


def test_bignSign2_batch_with_long_additional_input(orc, golden):
    eng = exp_engine()
    prm = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    from bee2_amd.engine import OID_BELT_HASH_DER
    n = 300
    hashes = orc.fill(32 * n, 7)
    privs = bytearray(orc.fill(32 * n, 8))
    for i in range(n):
        privs[32 * i + 31] &= 0x7F
        privs[32 * i] |= 1
    t = orc.fill(100, 9)                                      # > 64 octets: theta by the batched belt-hash (host staging vectors)
    sigs = ctypes.create_string_buffer(b"\xEE" * (48 * n), 48 * n)
    codes = (ctypes.c_uint32 * n)(*([0xEEEEEEEE] * n))

    def call():
        ctypes.memset(sigs, 0xEE, 48 * n)
        return eng.lib.bee2hip_bignSign2_batch(ctypes.byref(prm), bytes(OID_BELT_HASH_DER), _sz(len(OID_BELT_HASH_DER)), hashes, bytes(privs),
                                               t, _sz(len(t)), _sz(n), sigs, codes)

    def untouched():
        assert sigs.raw == b"\xEE" * (48 * n)
    failures, allocs = _walk(eng, call, untouched)
    assert allocs >= 3 and failures == min(allocs, 40)
    assert call() == 0 and all(c == 0 for c in codes)
    for i in (0, 1, n - 1):                                   # the signatures are the reference's (deterministic: bignSign2 with t)
        assert (0, sigs.raw[48 * i: 48 * i + 48]) == orc.sign2(128, bytes(OID_BELT_HASH_DER), hashes[32 * i: 32 * i + 32],
                                                                bytes(privs[32 * i: 32 * i + 32]), t)


def test_bignVerify_keyed_batch(orc, golden):
    eng = exp_engine()
    prm = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    from bee2_amd.engine import OID_BELT_HASH_DER
    hs, ss, ps = golden.bign_base_arrays()
    n = 512
    keys = sorted({ps[64 * i: 64 * i + 64] for i in range(n)})
    idx = (ctypes.c_uint32 * n)(*[keys.index(ps[64 * i: 64 * i + 64]) for i in range(n)])
    kb = b"".join(keys)
    sdam = bytearray(ss[: 48 * n]); sdam[48 * 5 + 3] ^= 1
    codes = (ctypes.c_uint32 * n)()
    eng.lib.bee2hip_internal_tune(21, 4)                      # a small table cache: every call rebuilds tables (vectors, maps, host threads)

    def call():
        for i in range(n):
            codes[i] = 0xEEEEEEEE
        return eng.lib.bee2hip_bignVerify_keyed_batch(ctypes.byref(prm), bytes(OID_BELT_HASH_DER), _sz(len(OID_BELT_HASH_DER)), hs[: 32 * n],
                                                      bytes(sdam), kb, _sz(len(keys)), idx, _sz(n), codes)

    def untouched():
        assert all(c == 0xEEEEEEEE for c in codes)
    want = orc.verify_batch(hs[: 32 * n], bytes(sdam), ps[: 64 * n])

    def check_ok():
        assert list(codes) == want
    try:
        failures, allocs = _walk(eng, call, untouched, check_ok=check_ok)
    finally:
        eng.lib.bee2hip_internal_tune(21, 1024)
    assert allocs >= 2 and failures >= 1          # (std::string / std::thread internals allocate inside libstdc++.so, out of the hook's reach)
    assert call() == 0 and list(codes) == want and codes[5] == 510
