"""CPU: the index arithmetic of the *_multi entry points (bee2_amd/csrc/multi.hip) -- callable
without a GPU -- against bee2_amd/shard.py and, with the ranges processed by the oracle, against the whole
batch processed at once (a mocked device count stands in for the GPUs)."""
import ctypes

import numpy as np
import pytest

import bee2_amd
from bee2_amd import shard

_sz = ctypes.c_size_t


@pytest.fixture(scope="module")
def lib():
    import os
    if not os.path.exists(bee2_amd.LIB_PATH):
        bee2_amd.build()
    L = ctypes.CDLL(bee2_amd.LIB_PATH)
    L.bee2hip_multi_plan.restype = ctypes.c_uint32
    return L


def plan(lib, n, parts, i):
    lo, cnt = _sz(0), _sz(0)
    assert lib.bee2hip_multi_plan(_sz(n), parts, i, ctypes.byref(lo), ctypes.byref(cnt)) == 0
    return lo.value, cnt.value


def test_plan_is_the_partition_shard_py_describes(lib):
    for n in (0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 2 ** 20, 2 ** 20 + 1, 2 ** 30 + 12345, 2 ** 40 + 7, 2 ** 63 + 11):
        for parts in (1, 2, 3, 4, 7, 8, 16):
            pos = 0
            sizes = []
            for i in range(parts):
                lo, cnt = plan(lib, n, parts, i)
                assert (lo, lo + cnt) == shard.shard_range(i, parts, n)
                assert lo == pos
                pos += cnt
                sizes.append(cnt)
            assert pos == n and max(sizes) - min(sizes) <= 1
    lo, cnt = _sz(0), _sz(0)
    for bad in ((10, 0, 0), (10, 4, 4), (10, 4, -1)):
        assert lib.bee2hip_multi_plan(_sz(bad[0]), bad[1], bad[2], ctypes.byref(lo), ctypes.byref(cnt)) == 109


@pytest.mark.parametrize("devices", [2, 3, 8])
def test_ctr_ranges_with_first_block_equal_the_whole_stream(lib, orc, golden, devices):
    """what bee2hip_beltCTR_bulk_multi does, with the oracle as the device: range i starts its counter at
    ctr0 + first_i; the pieces put together are the serially encrypted stream"""
    kw, c0 = orc.ctr_start(golden.H[128:160], golden.H[192:208])
    nblocks = 10_007
    data = np.frombuffer(orc.fill(16 * nblocks, 0xBE17), dtype=np.uint8).copy()
    whole = data.copy()
    orc.ctr_blocks_np(whole, kw, c0, first=0)
    pieces = data.copy()
    for i in range(devices):
        lo, cnt = plan(lib, nblocks, devices, i)
        part = pieces[16 * lo: 16 * (lo + cnt)]
        orc.ctr_blocks_np(part, kw, c0, first=lo)
    assert np.array_equal(pieces, whole)


@pytest.mark.parametrize("devices", [2, 5])
def test_independent_items_by_range_equal_the_whole_batch(lib, orc, golden, devices):
    n = 1003
    states = orc.fill(192 * n, 0xBA5F)
    whole = orc.bashF_batch(states)
    got = b""
    for i in range(devices):
        lo, cnt = plan(lib, n, devices, i)
        got += orc.bashF_batch(states[192 * lo: 192 * (lo + cnt)])
    assert got == whole
    hs, ss, ps = golden.bign_base_arrays()
    m = 200
    whole = orc.verify_batch(hs[:32 * m], ss[:48 * m], ps[:64 * m])
    got = []
    for i in range(devices):
        lo, cnt = plan(lib, m, devices, i)
        got += orc.verify_batch(hs[32 * lo: 32 * (lo + cnt)], ss[48 * lo: 48 * (lo + cnt)], ps[64 * lo: 64 * (lo + cnt)])
    assert got == whole
