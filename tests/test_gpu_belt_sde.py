"""-m gpu: SURVEY.md 8f-1 -- belt-sde over belt-wbl (mirrors test/crypto/belt_test.c:661-688; STB A.24-2 /
A.25-2 are in test_gpu_belt_modes.py::test_ecb_cbc_A9_A12_dropin via tests/golden/stb_kat.json)."""
import random

import pytest
import torch

from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu


def test_sde_golden_cases_one_shot_and_steps(golden):
    eng = engine()
    for c in golden.belt_sde:
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        code, e = eng.belt_mode("beltSDEEncr", msg, key, iv)
        assert code == 0 and e.hex() == c["sde_e"], c["blocks"]
        code, d = eng.belt_mode("beltSDEDecr", msg, key, iv)
        assert code == 0 and d.hex() == c["sde_d"], c["blocks"]
        assert eng.belt_mode("beltSDEDecr", e, key, iv)[1] == msg
        assert eng.belt_mode_steps("SDE", False, msg, key, iv, [len(msg)]).hex() == c["sde_e"]
        assert eng.belt_mode_steps("SDE", True, msg, key, iv, [len(msg)]).hex() == c["sde_d"]
    for bad in (b"", b"x" * 16, b"x" * 31, b"x" * 33):                  # belt_sde.c:79-80
        assert eng.belt_mode("beltSDEEncr", bad, b"k" * 32, b"i" * 16)[0] == 109
        assert eng.belt_mode("beltSDEDecr", bad, b"k" * 32, b"i" * 16)[0] == 109
    assert eng.belt_mode("beltSDEEncr", b"x" * 32, b"k" * 31, b"i" * 16)[0] == 109


def test_sde_one_state_many_sectors(orc, golden):
    """one beltSDEStart, then sectors of different sizes with different ivs through the same state"""
    eng = engine()
    rnd = random.Random(2)
    key = golden.H[128:160]
    import ctypes
    st = ctypes.create_string_buffer(eng.lib.beltSDE_keep())
    eng.lib.beltSDEStart(st, key, ctypes.c_size_t(32))
    for nb in (2, 3, 7, 32, 33, 100):
        msg, iv = rnd.randbytes(16 * nb), rnd.randbytes(16)
        b = ctypes.create_string_buffer(msg, len(msg))
        eng.lib.beltSDEStepE(b, ctypes.c_size_t(len(msg)), iv, st)
        assert b.raw[: len(msg)] == orc.sde(msg, key, iv)[1], nb
        eng.lib.beltSDEStepD(b, ctypes.c_size_t(len(msg)), iv, st)
        assert b.raw[: len(msg)] == msg, nb


@pytest.mark.parametrize("sector_bytes,nsectors", [(32, 1), (32, 1000), (48, 257), (512, 1), (512, 1025), (4096, 300),
                                                    (16 * 7, 64), (16 * 1000, 3)])
def test_sde_sectors_dev_vs_oracle(orc, golden, sector_bytes, nsectors):
    """the batch entry: every sector with its own iv, against the oracle sector by sector; then back"""
    eng = engine()
    key = golden.H[160:192]
    kw = bytes(orc.key_expand(key))
    data = orc.fill(sector_bytes * nsectors, sector_bytes + nsectors)
    ivs = orc.fill(16 * nsectors, 77 + nsectors)
    buf, div = dev(data), dev(ivs)
    eng.beltSDE_sectors_dev(0, buf, sector_bytes, kw, div)
    torch.cuda.synchronize()
    got = host(buf)
    step = max(1, nsectors // 40)                                        # the oracle is O(n^2) per sector: sample
    for i in list(range(0, nsectors, step)) + [nsectors - 1]:
        sec = data[i * sector_bytes:(i + 1) * sector_bytes]
        assert got[i * sector_bytes:(i + 1) * sector_bytes] == orc.sde(sec, key, ivs[16 * i:16 * i + 16])[1], i
    eng.beltSDE_sectors_dev(1, buf, sector_bytes, kw, div)
    torch.cuda.synchronize()
    assert host(buf) == data
    # decrypting fresh data equals the oracle's decryption as well
    buf2 = dev(data)
    eng.beltSDE_sectors_dev(1, buf2, sector_bytes, kw, div)
    torch.cuda.synchronize()
    got2 = host(buf2)
    for i in (0, nsectors // 2, nsectors - 1):
        sec = data[i * sector_bytes:(i + 1) * sector_bytes]
        assert got2[i * sector_bytes:(i + 1) * sector_bytes] == orc.sde(sec, key, ivs[16 * i:16 * i + 16], True)[1], i


def test_sde_large_batch_round_trip():
    """2^20 sectors of 512 bytes (512 MiB): D(E(x)) = x; a changed iv or a flipped bit changes the whole sector"""
    eng = engine()
    n, sb = 1 << 20, 512
    kw = bytes(range(32))
    x = torch.empty(n * sb, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(0x5DE)
    x.view(torch.int64).random_(generator=g)
    ivs = torch.empty(16 * n, dtype=torch.uint8, device="cuda")
    ivs.view(torch.int64).random_(generator=g)
    y = x.clone()
    eng.beltSDE_sectors_dev(0, y, sb, kw, ivs)
    torch.cuda.synchronize()
    assert not torch.equal(y[:sb], x[:sb])
    z = x.clone()
    z[5 * sb + 100] ^= 1                                                  # one bit in sector 5
    eng.beltSDE_sectors_dev(0, z, sb, kw, ivs)
    torch.cuda.synchronize()
    diff = (y != z).view(n, sb).sum(dim=1)
    assert int((diff > 0).sum()) == 1 and int(diff[5]) > sb * 0.9        # only sector 5, and almost every byte of it
    eng.beltSDE_sectors_dev(1, y, sb, kw, ivs)
    torch.cuda.synchronize()
    assert torch.equal(y, x)
