"""CPU tests of bee2_amd/csrc/host_small.hpp -- the drop-in layer's HOST path for small single calls (product code) --
against the oracle and the golden vectors.  The header is plain C++; tests/hostshim/host_small_shim.cpp gives it a
C view and this module builds that shim with g++ (no GPU, no HIP).  The same functions are reached on the GPU box
through libbee2hip.so's drop-in symbols with BEE2HIP_FORCE=cpu (tests/test_gpu_*.py run their fixtures both ways)."""
import ctypes
import os
import random
import subprocess

import pytest

from conftest import hostshim_san_flags

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hs(tmp_path_factory, orc):
    out = tmp_path_factory.mktemp("hostshim") / "libhostshim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror"] + hostshim_san_flags() + ["-o", str(out),
                           os.path.join(ROOT, "tests", "hostshim", "host_small_shim.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.hs_init(orc.beltH())
    return lib


def _sz(n):
    return ctypes.c_size_t(n)


def _kw(orc, key):
    return bytes(orc.key_expand(key))


def test_bashF_A2_and_random_states(hs, orc, golden):
    k = golden.kat["bashF_A2"]
    b = ctypes.create_string_buffer(bytes.fromhex(k["in"]), 192)
    hs.hs_bashF(b)
    assert b.raw.hex() == k["out"]
    rnd = random.Random(7)
    for _ in range(300):
        s = rnd.randbytes(192)
        b = ctypes.create_string_buffer(s, 192)
        hs.hs_bashF(b)
        assert b.raw == orc.bashF(s)


@pytest.mark.parametrize("l", [32, 128, 192, 256])
def test_sponge_absorb_equals_bashHash_for_any_split(hs, orc, l):
    rnd = random.Random(l)
    for n in (0, 1, 63, 64, 65, 127, 128, 129, 191, 192, 500, 4097):
        msg = rnd.randbytes(n)
        st = bytearray(192)
        st[184] = l // 4
        buf_len = 192 - l // 2
        pos = ctypes.c_size_t(0)
        s = (ctypes.c_ubyte * 192).from_buffer(st)
        off = 0
        while off < n:
            take = min(n - off, rnd.choice([1, 7, 64, 100, 1000]))
            hs.hs_sponge(s, _sz(buf_len), ctypes.byref(pos), msg[off:off + take], _sz(take))
            off += take
        fin = bytearray(st)
        for i in range(pos.value, buf_len):
            fin[i] = 0
        fin[pos.value] = 0x40
        f = (ctypes.c_ubyte * 192).from_buffer(fin)
        hs.hs_bashF(f)
        assert bytes(fin[: l // 4]) == orc.bashHash(l, msg)[1], (l, n)


def test_block_encr_decr(hs, orc, golden):
    H = orc.beltH()
    x = (ctypes.c_uint32 * 4).from_buffer_copy(H[:16])           # STB A.1
    hs.hs_encr(x, orc.key_expand(H[128:160]))
    assert bytes(x) == orc.block_encr(H[:16], H[128:160])
    rnd = random.Random(3)
    for klen in (16, 24, 32):
        for _ in range(200):
            key, blk = rnd.randbytes(klen), rnd.randbytes(16)
            x = (ctypes.c_uint32 * 4).from_buffer_copy(blk)
            hs.hs_encr(x, orc.key_expand(key))
            assert bytes(x) == orc.block_encr(blk, key)
            hs.hs_decr(x, orc.key_expand(key))
            assert bytes(x) == blk
            y = (ctypes.c_uint32 * 4).from_buffer_copy(blk)
            hs.hs_decr(y, orc.key_expand(key))
            assert bytes(y) == orc.block_decr(blk, key)


def test_ctr_blocks_with_state(hs, orc):
    rnd = random.Random(5)
    for n in (1, 15, 16, 17, 48, 100, 1000, 4099):
        key, iv, msg = rnd.randbytes(32), rnd.randbytes(16), rnd.randbytes(n)
        kw, c0 = orc.ctr_start(key, iv)
        ctr = (ctypes.c_uint32 * 4).from_buffer_copy(c0)
        if n == 100:                                             # a carry through the low words
            ctr = (ctypes.c_uint32 * 4)(0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFF, 5)
            # the oracle from the same counter: counter = E_K(iv) cannot be chosen, so compare blockwise
            buf = ctypes.create_string_buffer(msg, n)
            block, res = ctypes.create_string_buffer(16), ctypes.c_size_t(0)
            hs.hs_ctr(buf, _sz(n), kw, ctr, block, ctypes.byref(res))
            c = int.from_bytes(bytes((ctypes.c_uint32 * 4)(0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFF, 5)), "little")
            want = bytearray()
            for b in range((n + 15) // 16):
                c = (c + 1) % (1 << 128)
                g = orc.block_encr(c.to_bytes(16, "little"), key)
                want += bytes(a ^ b_ for a, b_ in zip(msg[16 * b: 16 * b + 16], g))
            assert buf.raw[:n] == bytes(want)
            assert int.from_bytes(bytes(ctr), "little") == c
            continue
        buf = ctypes.create_string_buffer(msg, n)
        block, res = ctypes.create_string_buffer(16), ctypes.c_size_t(0)
        hs.hs_ctr(buf, _sz(n), kw, ctr, block, ctypes.byref(res))
        assert buf.raw[:n] == orc.ctr(msg, key, iv)
        assert res.value == (16 - n % 16) % 16


def test_mac_any_split(hs, orc, golden):
    rnd = random.Random(11)
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 48, 100, 1000, 1024, 1025, 4096 + 16, 5000) * 3:
        key, msg = rnd.randbytes(32), rnd.randbytes(n)
        kw = _kw(orc, key)
        s, r, mac = ((ctypes.c_uint32 * 4)() for _ in range(3))
        block, filled = ctypes.create_string_buffer(16), ctypes.c_size_t(0)
        hs.hs_mac(kw, s, r, mac, block, ctypes.byref(filled), None, _sz(0), 1)
        off = 0
        while off < n:
            take = min(n - off, rnd.choice([1, 5, 16, 17, 32, 48, 64, 200, 1024]))   # whole blocks go straight from the caller's buffer
            hs.hs_mac(kw, s, r, mac, block, ctypes.byref(filled), msg[off:off + take], _sz(take), 2)
            off += take
        hs.hs_mac(kw, s, r, mac, block, ctypes.byref(filled), None, _sz(0), 4)
        assert bytes(mac)[:8] == orc.mac(msg, key), n


def test_belt_hash_stream(hs, orc):
    rnd = random.Random(13)
    H = orc.beltH()
    for n in (0, 1, 31, 32, 33, 64, 95, 1000, 4096):
        msg = rnd.randbytes(n)
        st = (ctypes.c_uint32 * 12).from_buffer_copy(H[:32] + bytes(16))
        full = n // 32
        hs.hs_hash(st, msg, _sz(full), 0, ctypes.c_uint64(0), ctypes.c_uint64(0))
        tail = msg[32 * full:]
        if tail:
            hs.hs_hash(st, tail + bytes(32 - len(tail)), _sz(1), 1, ctypes.c_uint64(8 * n), ctypes.c_uint64(0))
        else:
            hs.hs_hash(st, None, _sz(0), 1, ctypes.c_uint64(8 * n), ctypes.c_uint64(0))
        assert bytes(st)[:32] == orc.belt_hash(msg), n


def test_ecb_cbc_blocks(hs, orc):
    rnd = random.Random(17)
    for nb in (1, 2, 3, 17, 64):
        key, iv, msg = rnd.randbytes(32), rnd.randbytes(16), rnd.randbytes(16 * nb)
        kw = _kw(orc, key)
        ivw = (ctypes.c_uint32 * 4).from_buffer_copy(iv)
        for mode, want in ((0, orc.ecb(msg, key)[1]), (1, orc.ecb(msg, key, True)[1]), (2, orc.cbc(msg, key, iv, True)[1])):
            buf = ctypes.create_string_buffer(msg, len(msg))
            hs.hs_modes(mode, buf, _sz(nb), kw, ivw)
            assert buf.raw == want, (mode, nb)
        buf = ctypes.create_string_buffer(msg, len(msg))
        chain = ctypes.create_string_buffer(iv, 16)
        hs.hs_cbc_encr(buf, _sz(nb), kw, chain)
        assert buf.raw == orc.cbc(msg, key, iv)[1]
        assert chain.raw == buf.raw[-16:]


def test_bde_che_blocks_and_state(hs, orc):
    rnd = random.Random(19)
    for nb in (1, 2, 5, 64, 129):
        key, s0, msg = rnd.randbytes(32), rnd.randbytes(16), rnd.randbytes(16 * nb)
        kw = _kw(orc, key)
        for decr in (0, 1):
            buf = ctypes.create_string_buffer(msg, len(msg))
            s = (ctypes.c_uint32 * 4).from_buffer_copy(s0)
            hs.hs_bde(decr, buf, _sz(nb), kw, s)
            want, s_after = orc.bde_blocks_from(msg, key, s0, bool(decr))
            assert buf.raw == want and bytes(s) == s_after
        buf = ctypes.create_string_buffer(msg, len(msg))
        s = (ctypes.c_uint32 * 4).from_buffer_copy(s0)
        hs.hs_che(buf, _sz(nb), kw, s)
        want, s_after = orc.che_blocks_from(msg, key, s0)
        assert buf.raw == want and bytes(s) == s_after
    # the high bit of s set: the reduction polynomial comes in
    s0 = bytes(15) + b"\x80"
    buf = ctypes.create_string_buffer(bytes(32), 32)
    s = (ctypes.c_uint32 * 4).from_buffer_copy(s0)
    hs.hs_che(buf, _sz(2), _kw(orc, bytes(32)), s)
    assert (buf.raw, bytes(s)) == orc.che_blocks_from(bytes(32), bytes(32), s0)


def test_sde_sectors(hs, orc):
    rnd = random.Random(23)
    for nb in (2, 3, 4, 5, 7, 32, 33, 64, 256):
        key, iv, msg = rnd.randbytes(32), rnd.randbytes(16), rnd.randbytes(16 * nb)
        kw = _kw(orc, key)
        for decr in (0, 1):
            buf = ctypes.create_string_buffer(msg, len(msg))
            hs.hs_sde(decr, buf, _sz(len(msg)), iv, kw)
            code, want = orc.sde(msg, key, iv, bool(decr))
            assert code == 0 and buf.raw == want, (nb, decr)


def test_polyhash_against_python_gf128(hs):
    """t <- (t ^ X) * r in GF(2)[x] / (x^128 + x^7 + x^2 + x + 1), checked with Python integers as polynomials"""
    rnd = random.Random(29)
    MOD = (1 << 128) | 0x87

    def mul(a, b):
        r = 0
        for i in range(128):
            if (b >> i) & 1:
                r ^= a << i
        for i in range(r.bit_length() - 1, 127, -1):
            if (r >> i) & 1:
                r ^= MOD << (i - 128)
        return r
    # three forms of the product (host_small.hpp): 0 = what the library runs (the CPU's carry-less multiplier where it has
    # one), 1 = table of 16 multiples of r, 2 = the bit-serial definition; sparse / dense / boundary operands included
    specials = [0, 1, 2, 0x87, 1 << 63, 1 << 64, 1 << 127, (1 << 128) - 1, (1 << 127) | 1, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF << 64]
    cases = [(rnd.getrandbits(128), rnd.getrandbits(128), rnd.randbytes(n)) for n in (0, 1, 15, 16, 17, 32, 64, 65, 80, 100, 1024)]
    cases += [(a, b, bytes(16)) for a in specials for b in specials]
    cases += [(rnd.getrandbits(128), b, rnd.randbytes(160)) for b in specials]
    for t0, r0, data in cases:
        n = len(data)
        want = t0
        for off in range(0, n, 16):
            blk = data[off:off + 16]
            want = mul(want ^ int.from_bytes(blk + bytes(16 - len(blk)), "little"), r0)
        for form in (0, 1, 2):
            t = (ctypes.c_uint32 * 4).from_buffer_copy(t0.to_bytes(16, "little"))
            r = (ctypes.c_uint32 * 4).from_buffer_copy(r0.to_bytes(16, "little"))
            hs.hs_polyhash_form(t, r, data, _sz(n), form)
            assert int.from_bytes(bytes(t), "little") == want, (n, form, hex(t0), hex(r0))
    assert hs.hs_have_clmul() in (0, 1)
    # corner: the top bit of the accumulator set and r = x (a single shift with reduction)
    t = (ctypes.c_uint32 * 4).from_buffer_copy((1 << 127).to_bytes(16, "little"))
    r = (ctypes.c_uint32 * 4).from_buffer_copy((2).to_bytes(16, "little"))
    hs.hs_polyhash(t, r, bytes(16), _sz(16))
    assert int.from_bytes(bytes(t), "little") == 0x87
