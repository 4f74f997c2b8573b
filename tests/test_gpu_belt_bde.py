"""-m gpu: SURVEY.md 8f-1 -- belt-bde (mirrors test/crypto/belt_test.c:628-660; STB A.24-1 / A.25-1 are
in test_gpu_belt_modes.py::test_ecb_cbc_A9_A12_dropin via tests/golden/stb_kat.json)."""
import random

import pytest
import torch

from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu

POLY = (1 << 128) | 0x87


def gf_mul_xpow(s, e):
    """s * x^e in GF(2)[x] / (x^128 + x^7 + x^2 + x + 1), plain integer arithmetic (independent of
    both the kernels and the oracle)"""
    def mul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a <<= 1
            if a >> 128:
                a ^= POLY
            b >>= 1
        return r
    base, r = 2, s
    while e:
        if e & 1:
            r = mul(r, base)
        base = mul(base, base)
        e >>= 1
    return r


def test_bde_golden_cases_one_shot(golden):
    eng = engine()
    for c in golden.belt_bde:
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        code, e = eng.belt_mode("beltBDEEncr", msg, key, iv)
        assert code == 0 and e.hex() == c["bde_e"], c["blocks"]
        code, d = eng.belt_mode("beltBDEDecr", msg, key, iv)
        assert code == 0 and d.hex() == c["bde_d"], c["blocks"]
        assert eng.belt_mode("beltBDEDecr", e, key, iv)[1] == msg
    for bad in (b"", b"x" * 15, b"x" * 17, b"x" * 31):            # belt_bde.c:93-100
        assert eng.belt_mode("beltBDEEncr", bad, b"k" * 32, b"i" * 16)[0] == 109
        assert eng.belt_mode("beltBDEDecr", bad, b"k" * 32, b"i" * 16)[0] == 109
    assert eng.belt_mode("beltBDEEncr", b"x" * 16, b"k" * 31, b"i" * 16)[0] == 109


def test_bde_steps_carry_the_tweak(orc, golden):
    """any block-aligned split of the stream gives the one-shot result: the state's s is advanced
    exactly as the reference's loop does"""
    eng = engine()
    rnd = random.Random(11)
    for n in (16, 32, 48, 1024, 16 * 200, 16 * 1031):
        key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        msg = orc.fill(n, n)
        for decr in (False, True):
            want = orc.bde(msg, key, iv, decr)[1]
            for _ in range(3):
                splits, left = [], n
                while left > 16 and len(splits) < 12:
                    s = 16 * rnd.randrange(1, left // 16)
                    splits.append(s)
                    left -= s
                splits.append(left)
                assert eng.belt_mode_steps("BDE", decr, msg, key, iv, splits) == want, (n, decr, splits)


@pytest.mark.parametrize("nblocks", [1, 63, 64, 65, 1000, 8192, 8193, (1 << 17) + 3, 1 << 20, (1 << 20) + 77])
def test_bde_blocks_dev_vs_oracle(orc, golden, nblocks):
    eng = engine()
    key, iv = golden.H[128:160], golden.H[192:208]
    kw, s0 = eng.beltBDEStart(key, iv)
    assert kw == bytes(orc.key_expand(key))
    data = orc.fill(16 * nblocks, nblocks)
    want = orc.bde(data, key, iv)[1]
    src = dev(data)
    dst = torch.empty_like(src)
    sout = torch.zeros(16, dtype=torch.uint8, device="cuda")
    eng.beltBDE_blocks_dev(0, src, dst, kw, s0, 0, sout)
    torch.cuda.synchronize()
    assert host(dst) == want
    # the state after the piece: s0 * x^nblocks, checked with plain integer polynomial arithmetic
    assert int.from_bytes(host(sout), "little") == gf_mul_xpow(int.from_bytes(s0, "little"), nblocks)
    eng.beltBDE_blocks_dev(1, dst, dst, kw, s0)                 # decrypt in place restores the input
    torch.cuda.synchronize()
    assert host(dst) == data
    assert orc.bde(want, key, iv, True)[1] == data


def test_bde_stream_cut_into_pieces(orc, golden):
    """first_block: a stream processed in pieces (as ranks of a sharded job would) equals the whole"""
    eng = engine()
    rnd = random.Random(5)
    key, iv = golden.H[160:192], golden.H[208:224]
    kw, s0 = eng.beltBDEStart(key, iv)
    nblocks = 50_000
    data = orc.fill(16 * nblocks, 9)
    src = dev(data)
    for decr in (0, 1):
        want = orc.bde(data, key, iv, bool(decr))[1]
        for _ in range(3):
            cuts = sorted(set([0, nblocks] + [rnd.randrange(1, nblocks) for _ in range(6)] + [64, 65, 8192]))
            dst = torch.zeros_like(src)
            for a, b in zip(cuts, cuts[1:]):
                eng.beltBDE_blocks_dev(decr, src[16 * a:16 * b], dst[16 * a:16 * b], kw, s0, first_block=a)
            torch.cuda.synchronize()
            assert host(dst) == want, (decr, cuts)


def test_bde_jump_ahead_far_into_the_stream(orc, golden):
    """a piece that starts 2^40 + 5 (and 2^62) blocks in: the tweak kernel's square-and-multiply against
    integer arithmetic, and the piece against the same piece run from the jumped state"""
    eng = engine()
    key, iv = golden.H[128:160], golden.H[192:208]
    kw, s0 = eng.beltBDEStart(key, iv)
    data = orc.fill(16 * 3000, 4)
    src = dev(data)
    for first in (1, 63, 64, (1 << 32) - 1, (1 << 40) + 5, 1 << 62):
        sj = torch.zeros(16, dtype=torch.uint8, device="cuda")
        eng.beltBDE_blocks_dev(0, src[:0], src[:0], kw, s0, first_block=first, s_out=sj)   # empty piece: only the jump
        torch.cuda.synchronize()
        jumped = host(sj)
        assert int.from_bytes(jumped, "little") == gf_mul_xpow(int.from_bytes(s0, "little"), first), first
        a, b = torch.empty_like(src), torch.empty_like(src)
        eng.beltBDE_blocks_dev(0, src, a, kw, s0, first_block=first)
        eng.beltBDE_blocks_dev(0, src, b, kw, jumped, first_block=0)
        torch.cuda.synchronize()
        assert host(a) == host(b), first


def test_bde_large_round_trip_and_split_property():
    """BASELINE-sized stream (1 GiB): D(E(x)) = x and halves processed separately = whole"""
    eng = engine()
    n = (1 << 30) // 16
    kw, s0 = eng.beltBDEStart(bytes(range(32)), bytes(range(16)))
    x = torch.empty(16 * n, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(0xBDE)
    x.view(torch.int64).random_(generator=g)
    y = torch.empty_like(x)
    eng.beltBDE_blocks_dev(0, x, y, kw, s0)
    z = torch.empty_like(x)
    h = n // 2 + 12345
    eng.beltBDE_blocks_dev(0, x[:16 * h], z[:16 * h], kw, s0, first_block=0)
    eng.beltBDE_blocks_dev(0, x[16 * h:], z[16 * h:], kw, s0, first_block=h)
    torch.cuda.synchronize()
    assert torch.equal(y, z)
    assert not torch.equal(y[:1 << 20], x[:1 << 20])
    eng.beltBDE_blocks_dev(1, y, y, kw, s0)
    torch.cuda.synchronize()
    assert torch.equal(y, x)
