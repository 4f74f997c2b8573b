"""-m gpu: bashF batch kernel and the bash drop-in layer against the oracle and the
golden vectors (mirrors test/crypto/bash_test.c:41-154)."""
import os

import numpy as np
import pytest
import torch

from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu


def test_bashF_A2_in_slot0_and_golden_batch(golden):
    eng = engine()
    t = dev(golden.bashf_in)
    eng.bashF_batch_dev(t)
    torch.cuda.synchronize()
    out = host(t)
    assert out[:192].hex() == golden.kat["bashF_A2"]["out"]
    assert out == golden.bashf_out


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 127, 255, 256, 257, 1000, 4097])
def test_bashF_ragged_batch_sizes(orc, n):
    eng = engine()
    data = orc.fill(192 * n, 0xBA5F + n)
    guard = b"\xa5" * 192
    t = dev(data + guard)
    eng.bashF_batch_dev(t[: 192 * n])
    torch.cuda.synchronize()
    out = host(t)
    assert out[: 192 * n] == orc.bashF_batch(data)
    assert out[192 * n:] == guard                   # nothing written past the batch


def test_bashF_full_size_2pow20(orc):
    """BASELINE.json configs[1]: 2^20 independent states, every one compared"""
    eng = engine()
    n = 1 << 20
    h = np.empty(192 * n, dtype=np.uint8)
    orc.fill_np(h, 0xBA5F)
    t = torch.from_numpy(h).cuda()
    eng.bashF_batch_dev(t)
    torch.cuda.synchronize()
    want = h.copy()
    orc.bashF_batch_np(want, nthreads=os.cpu_count() or 1)
    assert np.array_equal(t.cpu().numpy(), want)


def test_bashF_host_pointer_api_and_dropin(orc, golden):
    eng = engine()
    data = orc.fill(192 * 300, 7)
    assert eng.bashF_batch(data) == orc.bashF_batch(data)
    assert eng.bashF(golden.H[:192]).hex() == golden.kat["bashF_A2"]["out"]


def test_bash_hash_A3_dropin(golden):
    eng = engine()
    for k in golden.kat["bash_hash"]:
        n = k["len"]
        code, d = eng.bashHash(k["l"], golden.H[:n])
        assert code == 0 and d.hex() == k["out"], k["name"]
        d2, ok = eng.bashHash_steps(k["l"], golden.H[:n], [n // 3, n - n // 3])
        assert d2.hex() == k["out"] and ok


def test_bash_hash_levels_and_splits_vs_oracle(orc):
    eng = engine()
    msg = orc.fill(1000, 42)
    for l in (16, 32, 64, 128, 144, 192, 256):
        for n in (0, 1, 63, 64, 65, 184, 191, 192, 193, 500, 1000):
            assert eng.bashHash(l, msg[:n]) == orc.bashHash(l, msg[:n])
    d, ok = eng.bashHash_steps(256, msg, [1, 62, 1, 64, 200, 672])
    assert d == orc.bashHash(256, msg)[1] and ok


def test_bash_hash_large_chunks_all_levels(orc):
    """chunks of >= 4 KiB take the 8-lanes-per-state kernel (rates of 8 .. 23 words: the rate reaches into
    rows 1 and 2 of the state for the small levels); one-shot, and split so that a partial block precedes
    a large chunk (byte-wise head, 8-lane bulk, byte-wise tail)"""
    eng = engine()
    msg = orc.fill(40_001, 7)
    for l in (16, 32, 64, 128, 144, 192, 256):
        want = orc.bashHash(l, msg)
        assert eng.bashHash(l, msg) == want, l
        assert eng.bashHash(l, msg[:4096]) == orc.bashHash(l, msg[:4096]), l
        for splits in ([40_001], [37, 9000, 30_964], [1, 4096 + 191, 35_713], [20_000, 20_001], [5000, 3, 34_998]):
            assert sum(splits) == len(msg)
            d, ok = eng.bashHash_steps(l, msg, splits)
            assert d == want[1] and ok, (l, splits)


def test_config0_bash256_1MiB_dropin(orc, golden):
    """BASELINE.json configs[0] shape through the drop-in bash256Hash path"""
    eng = engine()
    big = orc.fill(golden.big["len"], golden.big["seed"])
    code, d = eng.bashHash(128, big)
    assert code == 0 and d.hex() == golden.big["bash256"]


def test_bashF_batch_beyond_4GiB(orc):
    """Sized for the card: 2^26 states = 12 GiB in ONE launch (byte offsets past 2^32, lane indices past 2^24).  Every state
    is its own index (in its first 8 octets) over zeros; windows of 2^12 states at the start, either side of byte 2^32 and
    2^33, and at the end against the oracle; a guard state behind the batch stays untouched."""
    eng = engine()
    n = 1 << 26
    free, _ = torch.cuda.mem_get_info()
    if free < 192 * n + (2 << 30):
        pytest.skip("not enough HBM free for the 12 GiB case")
    t = torch.zeros(192 * (n + 1), dtype=torch.uint8, device="cuda")
    v = t.view(torch.int64)
    v[0:24 * n:24] = torch.arange(n, dtype=torch.int64, device="cuda")
    t[192 * n:] = 0xA5
    eng.bashF_batch_dev(t[: 192 * n])
    torch.cuda.synchronize()
    w = 1 << 12
    for first in (0, (1 << 32) // 192 - w // 2, (1 << 33) // 192 - w // 2, n // 2 + 12345, n - w):
        plain = np.zeros((w, 24), dtype=np.int64)
        plain[:, 0] = np.arange(first, first + w, dtype=np.int64)
        want = plain.view(np.uint8).reshape(-1).copy()
        orc.bashF_batch_np(want, nthreads=8)
        got = t[192 * first: 192 * (first + w)].cpu().numpy()
        assert np.array_equal(got, want), first
    assert bool((t[192 * n:] == 0xA5).all())
