"""-m gpu: SURVEY.md 8f-2 -- belt-dwp and belt-che (mirrors test/crypto/belt_test.c:473-563)."""
import random

import pytest
import torch

from gpulib import dev, engine, host
from test_oracle_golden import _dwp_ops_from_kat

pytestmark = pytest.mark.gpu

POLY = (1 << 128) | 0x87


def gf_mul(a, b):
    """GF(2)[x] / (x^128 + x^7 + x^2 + x + 1) on plain integers: independent of kernels and oracle"""
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        if a >> 128:
            a ^= POLY
        b >>= 1
    return r


def horner(t, r, data):
    """t <- (t ^ X) * r over the 16-byte blocks of data, the last one zero-padded (belt_dwp.c:96-101)"""
    for i in range(0, len(data), 16):
        t = gf_mul(t ^ int.from_bytes(data[i:i + 16].ljust(16, b"\0"), "little"), r)
    return t


MODES = pytest.mark.parametrize("mode", ["DWP", "CHE"])


def _gold(golden, mode):
    return golden.belt_dwp if mode == "DWP" else golden.belt_che


@MODES
def test_dwp_che_A19_A20_step_patterns(golden, mode):
    eng = engine()
    for k in _gold(golden, mode)["kat"]:
        key, iv = bytes.fromhex(k["key"]), bytes.fromhex(k["iv"])
        ops = _dwp_ops_from_kat(k) + [("V", bytes.fromhex(k["mac"])), ("V", bytes(8))]
        out, macs, oks = eng.dwp_steps(key, iv, ops, mode)
        assert out.hex() == k["out"] and macs[-1].hex() == k["mac"] and oks == [True, False], k["name"]
        crit, op = bytes.fromhex(k["crit"]), bytes.fromhex(k["open"])
        if k["op"] == "wrap":
            assert eng.dwp_wrap(crit, op, key, iv, mode) == (0, bytes.fromhex(k["out"]), bytes.fromhex(k["mac"]))
        else:
            assert eng.dwp_unwrap(crit, op, bytes.fromhex(k["mac"]), key, iv, mode) == (0, bytes.fromhex(k["out"]))


@MODES
def test_dwp_che_golden_short_and_long(orc, golden, mode):
    eng = engine()
    for c in _gold(golden, mode)["short"]:
        key, iv, crit, op = (bytes.fromhex(c[x]) for x in ("key", "iv", "crit", "open"))
        out, mac = bytes.fromhex(c["out"]), bytes.fromhex(c["mac"])
        assert eng.dwp_wrap(crit, op, key, iv, mode) == (0, out, mac), (len(crit), len(op))
        assert eng.dwp_unwrap(out, op, mac, key, iv, mode) == (0, crit)
        bad = bytes([mac[0] ^ 0x80]) + mac[1:]
        assert eng.dwp_unwrap(out, op, bad, key, iv, mode)[0] == 511                 # ERR_BAD_MAC
        if out:
            flipped = bytes([out[0] ^ 1]) + out[1:]
            assert eng.dwp_unwrap(flipped, op, mac, key, iv, mode)[0] == 511
    for c in _gold(golden, mode)["long"]:
        key, iv = bytes.fromhex(c["key"]), bytes.fromhex(c["iv"])
        crit, op = orc.fill(c["crit_len"], c["crit_seed"]), orc.fill(c["open_len"], c["open_seed"])
        code, out, mac = eng.dwp_wrap(crit, op, key, iv, mode)
        assert code == 0 and mac.hex() == c["mac"], (c["crit_len"], c["open_len"])
        assert orc.belt_hash(out).hex() == c["out_belt_hash"]
        assert eng.dwp_unwrap(out, op, mac, key, iv, mode) == (0, crit)
    assert eng.dwp_wrap(b"x", b"y", b"k" * 31, b"i" * 16, mode)[0] == 109


@MODES
def test_dwp_che_random_step_sequences_vs_oracle(orc, mode):
    """randomly cut Step{I,E,A} sequences with tags taken mid-stream: every tag and the ciphertext"""
    eng = engine()
    rnd = random.Random(23)

    def cut(b):
        parts = []
        while b:
            k = rnd.choice((1, 3, 7, 15, 16, 17, 33, 64, 1000))
            parts.append(b[:k])
            b = b[k:]
        return parts
    for _ in range(25):
        crit = rnd.randbytes(rnd.choice((0, 1, 15, 16, 17, 100, rnd.randrange(0, 5000))))
        op = rnd.randbytes(rnd.choice((0, 1, 15, 16, 17, rnd.randrange(0, 3000))))
        key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        ct = orc.dwp_wrap(crit, op, key, iv, mode)[1]
        ops = []
        for part in cut(op):
            ops.append(("I", part))
            if rnd.random() < 0.3:
                ops.append(("G",))
        ops += [("E", part) for part in cut(crit)]
        for part in cut(ct):
            ops.append(("A", part))
            if rnd.random() < 0.3:
                ops.append(("G",))
        ops.append(("G",))
        want_out, want_macs = orc.dwp_steps(key, iv, ops, mode)
        out, macs, _ = eng.dwp_steps(key, iv, ops, mode)
        assert out == want_out == ct and macs == want_macs, (len(crit), len(op))


@pytest.mark.parametrize("nbytes", [0, 1, 15, 16, 17, 16 * 63, 16 * 64, 16 * 65, 16 * 1023 + 9, 16 * 1024, 16 * 1025,
                                    16 * 2048 + 1, 16 * 65535, 16 * 65536 + 3, (1 << 22) + 5])
def test_dwp_absorb_dev_vs_integer_horner(orc, golden, nbytes):
    """the parallel Horner evaluation (chunks of >= 1024 blocks, 64 interleaved lanes, left-padded ragged
    chunk, finishing kernel) against the sequential definition on plain integers"""
    eng = engine()
    _, _, r, t0 = eng.beltDWPStart(golden.H[128:160], golden.H[192:208])
    data = orc.fill(nbytes, nbytes + 1)
    d = dev(data + bytes(16))
    tout = torch.zeros(16, dtype=torch.uint8, device="cuda")
    eng.beltDWP_absorb_dev(d, nbytes, r, t0, tout)
    torch.cuda.synchronize()
    want = horner(int.from_bytes(t0, "little"), int.from_bytes(r, "little"), data)
    assert int.from_bytes(host(tout), "little") == want


def test_dwp_absorb_large_split_property():
    """1 GiB: absorbing A then B from the intermediate t equals absorbing A || B (cut on a block boundary)"""
    eng = engine()
    _, _, r, t0 = eng.beltDWPStart(bytes(range(32)), bytes(range(16)))
    n = 1 << 30
    x = torch.empty(n, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(0xD3B)
    x.view(torch.int64).random_(generator=g)
    whole, mid, two = (torch.zeros(16, dtype=torch.uint8, device="cuda") for _ in range(3))
    eng.beltDWP_absorb_dev(x, n, r, t0, whole)
    cut = 16 * ((n // 32) + 12345)
    eng.beltDWP_absorb_dev(x, cut, r, t0, mid)
    torch.cuda.synchronize()
    eng.beltDWP_absorb_dev(x[cut:], n - cut, r, host(mid), two)
    torch.cuda.synchronize()
    assert host(whole) == host(two) and host(whole) != t0
    # one flipped bit anywhere changes the result
    x[n // 3] ^= 1
    eng.beltDWP_absorb_dev(x, n, r, t0, two)
    torch.cuda.synchronize()
    assert host(whole) != host(two)


# ----------------------------------------------------------------- belt-che keystream on the device
Q_INV = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF82            # (x + 1)^-1


def che_state(s0, j):
    """S_j of s <- s*x ^ 1 by the closed form S_0 x^j ^ (x^j ^ 1) / (x ^ 1), plain integers"""
    P, b, e = 1, 2, j
    while e:
        if e & 1:
            P = gf_mul(P, b)
        b = gf_mul(b, b)
        e >>= 1
    return gf_mul(s0, P) ^ gf_mul(P ^ 1, Q_INV)


def test_che_closed_form_matches_the_recurrence():
    assert gf_mul(3, Q_INV) == 1
    s = s0 = 0x0123456789ABCDEF0F1E2D3C4B5A6978
    for j in range(1, 200):
        s = (s << 1) ^ (POLY if s >> 127 else 0)
        s ^= 1
        assert s == che_state(s0, j)


@pytest.mark.parametrize("nblocks", [1, 63, 64, 65, 1000, 8192, 8193, (1 << 17) + 3, (1 << 20) + 77])
def test_che_blocks_dev_vs_oracle(orc, golden, nblocks):
    eng = engine()
    key, iv = golden.H[128:160], golden.H[192:208]
    kw, s0, _ = eng.beltCHEStart(key, iv)
    data = orc.fill(16 * nblocks, nblocks)
    want = orc.dwp_wrap(data, b"", key, iv, "CHE")[1]
    src = dev(data)
    dst = torch.empty_like(src)
    sout = torch.zeros(16, dtype=torch.uint8, device="cuda")
    eng.beltCHE_blocks_dev(src, dst, kw, s0, 0, sout)
    torch.cuda.synchronize()
    assert host(dst) == want
    assert int.from_bytes(host(sout), "little") == che_state(int.from_bytes(s0, "little"), nblocks)
    eng.beltCHE_blocks_dev(dst, dst, kw, s0)                    # the keystream again, in place: back to the input
    torch.cuda.synchronize()
    assert host(dst) == data


def test_che_stream_in_pieces_and_far_jumps(orc, golden):
    eng = engine()
    rnd = random.Random(6)
    key, iv = golden.H[160:192], golden.H[208:224]
    kw, s0, _ = eng.beltCHEStart(key, iv)
    nblocks = 40_000
    data = orc.fill(16 * nblocks, 3)
    want = orc.dwp_wrap(data, b"", key, iv, "CHE")[1]
    src = dev(data)
    for _ in range(3):
        cuts = sorted(set([0, nblocks] + [rnd.randrange(1, nblocks) for _ in range(6)] + [64, 65, 8192]))
        dst = torch.zeros_like(src)
        for a, b in zip(cuts, cuts[1:]):
            eng.beltCHE_blocks_dev(src[16 * a:16 * b], dst[16 * a:16 * b], kw, s0, first_block=a)
        torch.cuda.synchronize()
        assert host(dst) == want, cuts
    small = src[: 16 * 3000]
    for first in (1, 63, 64, (1 << 32) - 1, (1 << 40) + 5, 1 << 62):
        sj = torch.zeros(16, dtype=torch.uint8, device="cuda")
        eng.beltCHE_blocks_dev(small[:0], small[:0], kw, s0, first_block=first, s_out=sj)
        torch.cuda.synchronize()
        jumped = host(sj)
        assert int.from_bytes(jumped, "little") == che_state(int.from_bytes(s0, "little"), first), first
        a, b = torch.empty_like(small), torch.empty_like(small)
        eng.beltCHE_blocks_dev(small, a, kw, s0, first_block=first)
        eng.beltCHE_blocks_dev(small, b, kw, jumped, first_block=0)
        torch.cuda.synchronize()
        assert host(a) == host(b), first


def test_che_large_involution_and_split():
    eng = engine()
    n = (1 << 30) // 16
    kw, s0, _ = eng.beltCHEStart(bytes(range(32)), bytes(range(16)))
    x = torch.empty(16 * n, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(0xC4E)
    x.view(torch.int64).random_(generator=g)
    y, z = torch.empty_like(x), torch.empty_like(x)
    eng.beltCHE_blocks_dev(x, y, kw, s0)
    h = n // 2 + 4321
    eng.beltCHE_blocks_dev(x[:16 * h], z[:16 * h], kw, s0, first_block=0)
    eng.beltCHE_blocks_dev(x[16 * h:], z[16 * h:], kw, s0, first_block=h)
    torch.cuda.synchronize()
    assert torch.equal(y, z) and not torch.equal(y[:1 << 20], x[:1 << 20])
    eng.beltCHE_blocks_dev(y, y, kw, s0)
    torch.cuda.synchronize()
    assert torch.equal(y, x)
