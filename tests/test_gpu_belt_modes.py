"""-m gpu: SURVEY.md 8f-1 -- belt block decryption, ECB and CBC (mirrors
test/crypto/belt_test.c:178-215,288-396)."""
import random

import numpy as np
import pytest
import torch

from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu


def test_block_decr_A4_and_inverse(orc, golden):
    eng = engine()
    H = golden.H
    # A.4 (belt_test.c:208-214)
    assert eng.beltBlockDecr(H[64:80], H[160:192]).hex().upper() == "0DC5300600CAB840B38448E5E993F421"
    for i in range(8):
        blk, key = orc.fill(16, i), orc.fill(32, 100 + i)
        assert eng.beltBlockDecr(blk, key) == orc.block_decr(blk, key)
        assert eng.beltBlockDecr(eng.beltBlockEncr(blk, key), key) == blk


def test_ecb_cbc_A9_A12_dropin(golden):
    eng = engine()
    for k in golden.kat["belt_modes"]:
        msg, key = bytes.fromhex(k["in"]), bytes.fromhex(k["key"])
        iv = bytes.fromhex(k["iv"]) if k["iv"] else None
        code, out = eng.belt_mode(k["fn"], msg, key, iv)
        assert code == 0 and out.hex() == k["out"], k["name"]
        mode = k["fn"][4:7]                                   # ECB / CBC / BDE
        decr = k["fn"].endswith("Decr")
        # the reference's own split pattern: 16 or 32 bytes first, the rest (with the steal) second
        first = 32 if len(msg) == 48 and not decr else 16
        splits = [len(msg)] if mode == "SDE" else [first, len(msg) - first]     # belt-sde: one call = one sector
        out2 = eng.belt_mode_steps(mode, decr, msg, key, iv, splits)
        assert out2.hex() == k["out"], k["name"]


def test_ecb_cbc_golden_random_cases(golden):
    eng = engine()
    for c in golden.belt_bash:
        if "ecb_e" not in c:
            continue
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        assert eng.belt_mode("beltECBEncr", msg, key)[1].hex() == c["ecb_e"]
        assert eng.belt_mode("beltECBDecr", msg, key)[1].hex() == c["ecb_d"]
        assert eng.belt_mode("beltCBCEncr", msg, key, iv)[1].hex() == c["cbc_e"]
        assert eng.belt_mode("beltCBCDecr", msg, key, iv)[1].hex() == c["cbc_d"]
    assert eng.belt_mode("beltECBEncr", b"x" * 15, b"k" * 32)[0] == 109
    assert eng.belt_mode("beltCBCDecr", b"x" * 16, b"k" * 31, b"i" * 16)[0] == 109


def test_streaming_split_invariance(orc, golden):
    """block-aligned splits of the stream (bee2's precondition for all but the last call)"""
    eng = engine()
    rnd = random.Random(3)
    key, iv = golden.H[128:160], golden.H[192:208]
    for n in (16, 32, 47, 48, 100, 1000, 4096 + 5):
        msg = orc.fill(n, n)
        for decr in (False, True):
            want_e = orc.ecb(msg, key, decr)[1]
            want_c = orc.cbc(msg, key, iv, decr)[1]
            for _ in range(3):
                splits, left = [], n
                while left >= 48:
                    s = 16 * rnd.randrange(1, 1 + min(8, left // 16 - 2))
                    splits.append(s)
                    left -= s
                splits.append(left)
                assert eng.belt_mode_steps("ECB", decr, msg, key, None, splits) == want_e, (n, decr, splits)
                assert eng.belt_mode_steps("CBC", decr, msg, key, iv, splits) == want_c, (n, decr, splits)


@pytest.mark.parametrize("nblocks", [1, 63, 1024, 2049, 1 << 20])
def test_modes_blocks_dev_vs_oracle(orc, golden, nblocks):
    eng = engine()
    key, iv = golden.H[128:160], golden.H[192:208]
    kw = bytes(orc.key_expand(key))
    data = orc.fill(16 * nblocks, nblocks)
    src = dev(data)
    dst = torch.empty_like(src)
    eng.beltModes_blocks_dev(0, src, dst, kw)
    torch.cuda.synchronize()
    assert host(dst) == orc.ecb(data, key)[1]
    eng.beltModes_blocks_dev(1, dst, dst, kw)               # ECB decrypt in place restores the input
    torch.cuda.synchronize()
    assert host(dst) == data
    eng.beltModes_blocks_dev(2, src, dst, kw, iv)
    torch.cuda.synchronize()
    assert host(dst) == orc.cbc(data, key, iv, True)[1]


def test_cbc_encr_batch_lane_per_message(orc, golden):
    eng = engine()
    key = golden.H[128:160]
    kw = bytes(orc.key_expand(key))
    n, nblk = 300, 17
    msgs = orc.fill(n * nblk * 16, 5)
    ivs = orc.fill(n * 16, 6)
    dm, di = dev(msgs), dev(ivs)
    eng.beltCBCEncr_batch_dev(dm, nblk, kw, di)
    torch.cuda.synchronize()
    out, chain = host(dm), host(di)
    for m in (0, 1, 63, 64, 299):
        want = orc.cbc(msgs[m * nblk * 16:(m + 1) * nblk * 16], key, ivs[16 * m: 16 * m + 16])[1]
        assert out[m * nblk * 16:(m + 1) * nblk * 16] == want
        assert chain[16 * m: 16 * m + 16] == want[-16:]
