"""-m gpu: misuse of the device-pointer API is refused with an error code, not a GPU fault."""
import pytest
import torch

from bee2_amd.engine import EngineError
from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu


def test_misaligned_device_pointers_are_refused(orc, golden):
    eng = engine()
    kw, c0 = eng.beltCTRStart(golden.H[128:160], golden.H[192:208])
    bkw, bs0 = eng.beltBDEStart(golden.H[128:160], golden.H[192:208])
    _, _, r, t0 = eng.beltDWPStart(golden.H[128:160], golden.H[192:208])
    big = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    off = big[1:]                                               # data_ptr() + 1
    a16 = big[: 16 * 64]
    tout = torch.zeros(16, dtype=torch.uint8, device="cuda")
    hs, ss, ps = golden.bign_base_arrays()
    dh, ds, dp = dev(hs[:32 * 4]), dev(ss[:48 * 4]), dev(ps[:64 * 4])
    codes = torch.zeros(4, dtype=torch.int32, device="cuda")
    calls = {
        "bashF": lambda: eng.bashF_batch_dev(off[: 192 * 4]),
        "ctr": lambda: eng.beltCTR_blocks_dev(off[: 16 * 64], kw, c0, 0),
        "ecb src": lambda: eng.beltModes_blocks_dev(0, off[: 16 * 64], a16, kw),
        "ecb dst": lambda: eng.beltModes_blocks_dev(0, a16, off[: 16 * 64], kw),
        "bde": lambda: eng.beltBDE_blocks_dev(0, off[: 16 * 64], a16, bkw, bs0),
        "che": lambda: eng.beltCHE_blocks_dev(a16, off[: 16 * 64], bkw, bs0),
        "sde": lambda: eng.beltSDE_sectors_dev(0, off[: 512 * 2], 512, bkw, big[:32]),
        "sde ivs": lambda: eng.beltSDE_sectors_dev(0, big[: 512 * 2], 512, bkw, off[:32]),
        "dwp absorb": lambda: eng.beltDWP_absorb_dev(off, 1000, r, t0, tout),
        "verify": lambda: eng.bign128Verify_batch_dev(dh.new_zeros(32 * 4 + 1)[1:], ds, dp, codes),
        "fused": lambda: eng.bashHash_beltMAC_batch_dev(off[: 64 * 8], 64, 256, golden.H[128:160],
                                                        torch.zeros(64 * 8, dtype=torch.uint8, device="cuda"),
                                                        torch.zeros(8 * 8, dtype=torch.uint8, device="cuda"), n=8),
    }
    for name, call in calls.items():
        with pytest.raises(EngineError) as e:
            call()
        assert "err 109" in str(e.value), (name, str(e.value))
    # bad sector size / unknown flags
    with pytest.raises(EngineError):
        eng.beltSDE_sectors_dev(0, big[: 16 * 2], 16, bkw, big[:32])          # one block is not a sector
    with pytest.raises(EngineError):
        eng.beltSDE_sectors_dev(0, big[: 40 * 2], 40, bkw, big[:32])          # not whole blocks
    # ... and the device is still healthy: a correct call right after
    data = orc.fill(16 * 100, 1)
    buf = dev(data)
    eng.beltCTR_blocks_dev(buf, kw, c0, 0)
    torch.cuda.synchronize()
    assert host(buf) == orc.ctr(data, golden.H[128:160], golden.H[192:208])


def test_absurd_sizes_fail_with_out_of_memory_not_a_crash():
    """the device buffer is allocated before the host pointer is read: a size no GPU has is ERR_OUTOFMEMORY"""
    import ctypes
    eng = engine()
    buf = ctypes.create_string_buffer(192)
    huge = ctypes.c_size_t(1 << 42)                                   # 2^42 states = 768 TiB
    assert eng.lib.bee2hip_bashF_batch(buf, huge) == 110
    st = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
    eng.lib.beltCTRStart(st, bytes(32), ctypes.c_size_t(32), bytes(16))
    assert eng.lib.bee2hip_beltCTR_bulk(buf, ctypes.c_size_t(1 << 50), st) in (110, 109)
    # and the library still works afterwards
    code, out = eng.beltCTR(b"abc", bytes(32), bytes(16))
    assert code == 0 and len(out) == 3


def test_states_can_be_cloned_mid_stream(orc, golden):
    """bee2: "states are flat PODs of X_keep() bytes that may be memcpy-cloned" (SURVEY.md 8b Ownership).
    Feed a common prefix, copy the state bytes, continue the two copies with different data."""
    import ctypes
    eng = engine()
    L = eng.lib
    sz = ctypes.c_size_t
    key, iv = golden.H[128:160], golden.H[192:208]
    pre, a, b = orc.fill(1000 + 7, 1), orc.fill(5000 + 3, 2), orc.fill(333, 3)

    def clone(st):
        return ctypes.create_string_buffer(st.raw, len(st.raw))
    # beltCTR
    st = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(st, key, sz(32), iv)
    p = ctypes.create_string_buffer(pre, len(pre))
    L.beltCTRStepE(p, sz(len(pre)), st)
    for tail in (a, b):
        c = clone(st)
        t = ctypes.create_string_buffer(tail, len(tail))
        L.beltCTRStepE(t, sz(len(tail)), c)
        assert p.raw[: len(pre)] + t.raw[: len(tail)] == orc.ctr(pre + tail, key, iv)
    # beltMAC, bashHash, beltHash: digest of prefix || tail from a cloned state
    st = ctypes.create_string_buffer(L.beltMAC_keep())
    L.beltMACStart(st, key, sz(32))
    L.beltMACStepA(pre, sz(len(pre)), st)
    for tail in (a, b):
        c = clone(st)
        L.beltMACStepA(tail, sz(len(tail)), c)
        m = ctypes.create_string_buffer(8)
        L.beltMACStepG(m, c)
        assert m.raw == orc.mac(pre + tail, key)
    st = ctypes.create_string_buffer(L.bashHash_keep())
    L.bashHashStart(st, sz(128))
    L.bashHashStepH(pre, sz(len(pre)), st)
    for tail in (a, b):
        c = clone(st)
        L.bashHashStepH(tail, sz(len(tail)), c)
        d = ctypes.create_string_buffer(32)
        L.bashHashStepG(d, sz(32), c)
        assert d.raw == orc.bashHash(128, pre + tail)[1]
    st = ctypes.create_string_buffer(L.beltHash_keep())
    L.beltHashStart(st)
    L.beltHashStepH(pre, sz(len(pre)), st)
    for tail in (a, b):
        c = clone(st)
        L.beltHashStepH(tail, sz(len(tail)), c)
        d = ctypes.create_string_buffer(32)
        L.beltHashStepG(d, c)
        assert d.raw == orc.belt_hash(pre + tail)
    # belt-dwp / belt-che: open data in the common prefix, different critical data afterwards
    for mode in ("DWP", "CHE"):
        f = lambda n: getattr(L, f"belt{mode}{n}")
        st = ctypes.create_string_buffer(f("_keep")())
        f("Start")(st, key, sz(32), iv)
        f("StepI")(pre, sz(len(pre)), st)
        for tail in (a, b):
            c = clone(st)
            t = ctypes.create_string_buffer(tail, len(tail))
            f("StepE")(t, sz(len(tail)), c)
            f("StepA")(t, sz(len(tail)), c)
            m = ctypes.create_string_buffer(8)
            f("StepG")(m, c)
            assert (0, t.raw[: len(tail)], m.raw) == orc.dwp_wrap(tail, pre, key, iv, mode), mode
    # belt-bde: whole blocks
    pre16, a16, b16 = pre[:992], a[:4992], b[:320]
    st = ctypes.create_string_buffer(L.beltBDE_keep())
    L.beltBDEStart(st, key, sz(32), iv)
    p = ctypes.create_string_buffer(pre16, len(pre16))
    L.beltBDEStepE(p, sz(len(pre16)), st)
    for tail in (a16, b16):
        c = clone(st)
        t = ctypes.create_string_buffer(tail, len(tail))
        L.beltBDEStepE(t, sz(len(tail)), c)
        assert p.raw[: len(pre16)] + t.raw[: len(tail)] == orc.bde(pre16 + tail, key, iv)[1]
