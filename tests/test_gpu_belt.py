"""-m gpu: belt block / CTR kernels and the belt drop-in layer (mirrors
test/crypto/belt_test.c:178-215,423-472)."""
import ctypes
import random

import numpy as np
import pytest
import torch

from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu


def test_belt_block_A1_dropin(golden):
    eng = engine()
    k = golden.kat["belt_block_A1"]
    assert eng.beltBlockEncr(bytes.fromhex(k["in"]), bytes.fromhex(k["key"])).hex() == k["out"]


def test_belt_block_encr_many(orc, golden):
    eng = engine()
    key = golden.H[128:160]
    kw = orc.key_expand(key)
    data = orc.fill(16 * 1000, 3)
    t = dev(data)
    eng.beltBlockEncr_dev(t, bytes(kw))
    torch.cuda.synchronize()
    out = host(t)
    for i in range(1000):
        assert out[16 * i: 16 * i + 16] == orc.block_encr(data[16 * i: 16 * i + 16], key), i


def test_belt_ctr_A15_A16_dropin_with_state(orc, golden):
    eng = engine()
    for k in golden.kat["belt_ctr"]:
        msg, key, iv = (bytes.fromhex(k[x]) for x in ("in", "key", "iv"))
        ct, st = eng.beltCTR_steps(msg, key, iv, k["splits"])
        assert ct.hex() == k["out"], k["name"]
        # the streaming state must end up exactly as beltCTRStepE leaves it
        ost = ctypes.create_string_buffer(72)
        orc.lib.orc_beltCTRStart(ost, key, ctypes.c_size_t(len(key)), iv)
        buf = ctypes.create_string_buffer(msg, len(msg))
        off = 0
        for s in k["splits"]:
            orc.lib.orc_beltCTRStepE(ctypes.byref(buf, off), ctypes.c_size_t(s), ost)
            off += s
        assert st == ost.raw
        code, one = eng.beltCTR(msg, key, iv)
        assert code == 0 and one.hex() == k["out"]


def test_belt_ctr_golden_random_cases(golden):
    eng = engine()
    for c in golden.belt_bash:
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        ct, _ = eng.beltCTR_steps(msg, key, iv, c["splits"])
        assert ct.hex() == c["ctr"]
        assert eng.beltCTR(msg, key, iv)[1].hex() == c["ctr"]


@pytest.mark.parametrize("nblocks", [1, 2, 63, 64, 1023, 1024, 2047, 2048, 2049, 100_000, 1 << 20])
def test_belt_ctr_blocks_dev_vs_oracle(orc, golden, nblocks):
    eng = engine()
    kw, c0 = orc.ctr_start(golden.H[128:160], golden.H[192:208])
    data = np.frombuffer(orc.fill(16 * nblocks, nblocks), dtype=np.uint8).copy()
    t = torch.from_numpy(data.copy()).cuda()
    first = 12345
    eng.beltCTR_blocks_dev(t, kw, c0, first)
    torch.cuda.synchronize()
    want = data.copy()
    orc.ctr_blocks_np(want, kw, c0, first=first, nthreads=8)
    assert np.array_equal(t.cpu().numpy(), want)


def test_belt_ctr_counter_carries(orc, golden):
    """128-bit little-endian counter: carries across every 32-bit word (belt_ctr.c:27-35)"""
    eng = engine()
    kw = bytes(orc.key_expand(golden.H[128:160]))
    # (round 3: the kernel hoists the G-box of the counter's upper half out of the block loop unless the lower 64 bits wrap
    # inside the launch -- the last two pairs put the wrap on the launch's last block and one block beyond it)
    for c0_int in (2 ** 32 - 3, 2 ** 64 - 3, 2 ** 96 - 3, 2 ** 128 - 3, 2 ** 64 - 1 - 2 ** 20,
                   2 ** 64 - 4096, 2 ** 64 - 4097, 7 * 2 ** 64 + 2 ** 64 - 4096, 7 * 2 ** 64 + 2 ** 64 - 4097, 2 ** 128 - 4096):
        c0 = c0_int.to_bytes(16, "little")
        for first in (0, 2 ** 33, 2 ** 64 - 10):
            data = np.frombuffer(orc.fill(16 * 4096, 9), dtype=np.uint8).copy()
            t = torch.from_numpy(data.copy()).cuda()
            eng.beltCTR_blocks_dev(t, kw, c0, first)
            torch.cuda.synchronize()
            want = data.copy()
            orc.ctr_blocks_np(want, kw, c0, first=first)
            assert np.array_equal(t.cpu().numpy(), want), (hex(c0_int), first)


def test_belt_ctr_streaming_split_invariance(orc, golden):
    """any split of the stream gives the same bytes and the same final state"""
    eng = engine()
    rnd = random.Random(11)
    key, iv = golden.H[128:160], golden.H[192:208]
    msg = orc.fill(5000, 5)
    want = orc.ctr(msg, key, iv)
    for _ in range(6):
        splits, left = [], len(msg)
        while left:
            s = min(left, rnd.choice((1, 5, 15, 16, 17, 31, 33, 64, 255, 1000)))
            splits.append(s)
            left -= s
        ct, _ = eng.beltCTR_steps(msg, key, iv, splits)
        assert ct == want


def test_belt_ctr_full_size_16GiB(orc, golden):
    """BASELINE.json configs[2]: 16 GiB stream, one key.  EVERY block is compared with the oracle (all host cores,
    256 MiB at a time: the plaintext of a chunk is regenerated from its seed, so nothing but the chunk under test
    sits in host memory), then the size-independent property E(E(x)) = x over the whole buffer."""
    import os
    eng = engine()
    nbytes = 16 << 30
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes + (2 << 30):
        pytest.skip("not enough HBM free for the 16 GiB case")
    kw, c0 = orc.ctr_start(golden.H[128:160], golden.H[192:208])
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    chunk = 256 << 20
    gen = torch.Generator(device="cuda")

    def plain(i, out):                                 # chunk i of the synthetic stream, generated in HBM
        gen.manual_seed(0xBE17 + i)
        out.view(torch.int64).random_(generator=gen)
    for i in range(nbytes // chunk):
        plain(i, buf[i * chunk:(i + 1) * chunk])
    checksum0 = int(buf.view(torch.int64).sum().item())
    eng.beltCTR_blocks_dev(buf, kw, c0, 0)
    torch.cuda.synchronize()
    tmp = torch.empty(chunk, dtype=torch.uint8, device="cuda")
    threads = os.cpu_count() or 8
    for i in range(nbytes // chunk):
        plain(i, tmp)
        want = tmp.cpu().numpy()
        orc.ctr_blocks_np(want, kw, c0, first=i * (chunk // 16), nthreads=threads)
        got = buf[i * chunk:(i + 1) * chunk].cpu().numpy()
        assert np.array_equal(got, want), f"chunk {i}"
    assert int(buf.view(torch.int64).sum().item()) != checksum0
    eng.beltCTR_blocks_dev(buf, kw, c0, 0)             # decrypt = encrypt
    torch.cuda.synchronize()
    assert int(buf.view(torch.int64).sum().item()) == checksum0
    plain(0, tmp)
    assert torch.equal(buf[:chunk], tmp)
    plain(nbytes // chunk - 1, tmp)
    assert torch.equal(buf[nbytes - chunk:], tmp)


def test_belt_mac_A17_and_golden_dropin(orc, golden):
    eng = engine()
    for k in golden.kat["belt_mac"]:
        msg, key = bytes.fromhex(k["in"]), bytes.fromhex(k["key"])
        code, tag = eng.beltMAC(msg, key)
        assert code == 0 and tag.hex() == k["out"]
        tag2, ok = eng.beltMAC_steps(msg, key, [len(msg) // 2, len(msg) - len(msg) // 2])
        assert tag2.hex() == k["out"] and ok
    for c in golden.belt_bash:
        msg, key = bytes.fromhex(c["msg"]), bytes.fromhex(c["key"])
        assert eng.beltMAC(msg, key)[1].hex() == c["mac"]
        tag, ok = eng.beltMAC_steps(msg, key, c["splits"])
        assert tag.hex() == c["mac"] and ok


def test_belt_ctr_and_ecb_streams_beyond_2_32_blocks(orc, golden):
    """Sized for the card (288 GB of HBM): ONE launch over an 80 GiB stream = 5 * 2^30 blocks, i.e. block indices and byte
    offsets past 2^32 / 2^36.  The plaintext is zero, so the CTR ciphertext is the gamma: windows of 1 MiB at the start, either
    side of block 2^32, at 3/4 of the stream and at its end against the oracle (which jumps to any block index), then decryption
    of the whole stream back to zero.  The same buffer through the ECB kernel pair: E at the windows against the oracle, D(E(x)) = x."""
    import os
    eng = engine()
    nbytes = 80 << 30
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes + (4 << 30):
        pytest.skip("not enough HBM free for the 80 GiB case")
    kw, c0 = orc.ctr_start(golden.H[128:160], golden.H[192:208])
    buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    eng.beltCTR_blocks_dev(buf, kw, c0, 0)
    torch.cuda.synchronize()
    win = 1 << 20
    threads = min(os.cpu_count() or 8, 32)
    starts = [0, (1 << 36) - win, 1 << 36, (1 << 36) + (1 << 33) + 5 * win, 60 << 30, nbytes - win]
    for s in starts:
        want = np.zeros(win, dtype=np.uint8)
        orc.ctr_blocks_np(want, kw, c0, first=s // 16, nthreads=threads)
        assert np.array_equal(buf[s:s + win].cpu().numpy(), want), f"CTR window at byte {s}"
    eng.beltCTR_blocks_dev(buf, kw, c0, 0)
    torch.cuda.synchronize()
    assert int(buf.view(torch.int64).count_nonzero().item()) == 0
    # ECB over the same 5 * 2^30 blocks: a plaintext that differs per block (the block index in its first 8 octets)
    v = buf.view(torch.int64)
    step = 1 << 28
    for a in range(0, v.numel(), step):                       # even positions = block index, odd positions stay 0
        b = min(a + step, v.numel())
        v[a:b:2] = torch.arange(a // 2, (b + 1) // 2, dtype=torch.int64, device="cuda")
    key = golden.H[128:160]
    eng.beltModes_blocks_dev(0, buf, buf, kw)
    torch.cuda.synchronize()
    for s in starts:
        idx = np.arange(s // 16, (s + win) // 16, dtype=np.int64)
        plain = np.zeros((win // 16, 2), dtype=np.int64)
        plain[:, 0] = idx
        code, want = orc.ecb(plain.tobytes(), key)
        assert code == 0 and buf[s:s + win].cpu().numpy().tobytes() == want, f"ECB window at byte {s}"
    eng.beltModes_blocks_dev(1, buf, buf, kw)
    torch.cuda.synchronize()
    for s in starts:
        got = buf[s:s + win].cpu().numpy().view(np.int64).reshape(-1, 2)
        assert np.array_equal(got[:, 0], np.arange(s // 16, (s + win) // 16, dtype=np.int64)) and not got[:, 1].any(), s
    assert int(v[1::2].count_nonzero().item()) == 0
