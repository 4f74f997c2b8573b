"""CPU tests of bee2_amd/csrc/host_bign.hpp -- the drop-in layer's HOST path for one signature verification (product
code) -- against the reference's verdicts in the committed fixtures and against the oracle (verdict AND <x_R>).
tests/hostshim/host_bign_shim.cpp gives the header a C view; this module builds it with g++ (no GPU, no HIP).  The same
code is reached on the GPU box through bignVerify / bign128Verify / bign192Verify / bign256Verify of libbee2hip.so
(tests/test_gpu_hostpath.py, tests/test_gpu_bign.py run their drop-in fixtures through both paths)."""
import ctypes
import os
import random
import subprocess

import pytest

from conftest import hostshim_san_flags

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = {128: 189, 192: 317, 256: 569}


@pytest.fixture(scope="module")
def hb(tmp_path_factory, orc):
    out = tmp_path_factory.mktemp("hostshim") / "libhostbign.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror"] + hostshim_san_flags() + ["-o", str(out),
                           os.path.join(ROOT, "tests", "hostshim", "host_bign_shim.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.hb_verify.restype = ctypes.c_uint32
    lib.hb_pubkey_val.restype = ctypes.c_uint32
    lib.hb_init(orc.beltH())
    return lib


def _verify(hb, l, oid, h, s, p, want_rx=False):
    rx = ctypes.create_string_buffer(l // 4)
    code = hb.hb_verify(ctypes.c_size_t(l), bytes(oid), ctypes.c_size_t(len(oid)), bytes(h), bytes(s), bytes(p), rx)
    return (code, rx.raw) if want_rx else code


def _field(hb, l, op, a, b=0):
    n = l // 4
    r = ctypes.create_string_buffer(n)
    hb.hb_field(ctypes.c_size_t(l), op, r, a.to_bytes(n, "little"), b.to_bytes(n, "little"))
    return int.from_bytes(r.raw, "little")


@pytest.mark.parametrize("l", [128, 192, 256])
def test_field_ops_against_python_integers(hb, l):
    p = 2 ** (2 * l) - C[l]
    rnd = random.Random(l)
    corner = [0, 1, 2, C[l], p - 1, p - 2, p - C[l], 2 ** (2 * l - 1), 2 ** 64 - 1, 2 ** (2 * l - 64), (2 ** (2 * l) - 1) % p,
              p - 2 ** 64, 2 ** 64, 2 ** 128 - 1]
    vals = corner + [rnd.randrange(p) for _ in range(200)]
    for a in vals:
        assert _field(hb, l, 1, a) == a * a % p
        if a:
            assert _field(hb, l, 4, a) * a % p == 1
        for b in corner + [rnd.randrange(p) for _ in range(6)]:
            assert _field(hb, l, 0, a, b) == a * b % p
            assert _field(hb, l, 2, a, b) == (a + b) % p
            assert _field(hb, l, 3, a, b) == (a - b) % p
    assert _field(hb, l, 4, 0) == 0


def test_wnaf_digits_recompose(hb):
    rnd = random.Random(5)
    for w, bits in ((5, 129), (5, 193), (5, 257), (7, 128), (7, 192), (7, 256)):
        nl = (bits + 63) // 64
        for k in [0, 1, 2, 2 ** bits - 1, 2 ** (bits - 1), 2 ** (bits - 1) + 1] + [rnd.getrandbits(bits) for _ in range(100)]:
            out = (ctypes.c_int8 * (64 * nl + 2))()
            n = hb.hb_wnaf(out, k.to_bytes(8 * nl, "little"), nl, w)
            d = list(out[:n])
            assert n <= bits + 1
            assert sum(x << i for i, x in enumerate(d)) == k
            assert all(x == 0 or (x % 2 and abs(x) < 2 ** (w - 1)) for x in d)
            assert all(not (d[i] and any(d[i + 1:i + w])) for i in range(n))


def test_G2_G3_and_the_edge_cases_of_the_256_bit_curve(hb, orc, golden):
    from bee2_amd.engine import LEVEL_OID
    for k in golden.kat["bign_verify"]:               # bign_test.c:338-357,388-400
        assert _verify(hb, 128, LEVEL_OID[128], *(bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey"))) == k["code"], k["name"]
    for k in golden.bign_edge:                         # the reference's codes, incl. P = +-Q inside the addition, R = O
        h, s, p = (bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey"))
        code, rx = _verify(hb, 128, LEVEL_OID[128], h, s, p, True)
        assert code == k["code"], k["name"]
        ocode, orx = orc.verify_rx(h, s, p)
        assert ocode == code
        if any(orx) and orc.pubkey_val(128, p) == 0:      # off the curve R depends on the addition chain; the verdict does not
            assert rx == orx, k["name"]


def test_genuine_and_corrupted_signatures_256_bit_curve(hb, orc, golden):
    from bee2_amd.engine import LEVEL_OID
    rnd = random.Random(11)
    for h, s, p in golden.bign_base[:300]:
        assert _verify(hb, 128, LEVEL_OID[128], h, s, p) == 0
        which = rnd.randrange(3)
        bad = [bytearray(h), bytearray(s), bytearray(p)]
        bad[which][rnd.randrange(len(bad[which]))] ^= 1 << rnd.randrange(8)
        assert _verify(hb, 128, LEVEL_OID[128], *bad) == orc.verify(*(bytes(x) for x in bad))


@pytest.mark.parametrize("l", [192, 256])
def test_wider_curves(hb, orc, golden, l):
    from bee2_amd.engine import LEVEL_OID
    d = golden.bign_big[str(l)]
    for t in d["base"][:64]:
        assert _verify(hb, l, LEVEL_OID[l], *(bytes.fromhex(t[x]) for x in ("hash", "sig", "pubkey"))) == 0
    for e in d["edge"]:
        assert _verify(hb, l, LEVEL_OID[l], *(bytes.fromhex(e[x]) for x in ("hash", "sig", "pubkey"))) == e["code"], (l, e["name"])
    rnd = random.Random(l)
    for t in d["base"][64:128]:
        h, s, p = (bytearray.fromhex(t[x]) for x in ("hash", "sig", "pubkey"))
        which = (h, s, p)[rnd.randrange(3)]
        which[rnd.randrange(len(which))] ^= 1 << rnd.randrange(8)
        assert _verify(hb, l, LEVEL_OID[l], h, s, p) == orc.verify_l(l, LEVEL_OID[l], h, s, p)


def test_other_oids_of_every_length(hb, golden):
    for c in golden.bign_oid_lengths + golden.bign_oid_long:
        got = _verify(hb, int(c["l"]), bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["sig"]),
                      bytes.fromhex(c["pubkey"]))
        assert got == c["code"], (c["l"], len(c["oid"]) // 2)


def test_sigvfy_pipeline_verdicts(hb, golden):
    from bee2_amd.engine import LEVEL_OID
    for l in (128, 192, 256):
        for it in golden.sigvfy_pipeline[str(l)]:
            pub, sig, dig = (bytes.fromhex(it[k]) for k in ("pubkey", "sig", "digest"))
            assert _verify(hb, l, LEVEL_OID[l], dig, sig, pub) == it["verify"]


@pytest.mark.parametrize("l", [128, 192, 256])
def test_pubkey_val_reference_codes(hb, golden, l):
    """bignPubkeyVal (bign_misc.c:319-365) on the reference-generated cases"""
    cases = golden.bign_pubkey_val[str(l)]
    bad = [(c["name"], c["code"]) for c in cases
           if hb.hb_pubkey_val(ctypes.c_size_t(l), bytes.fromhex(c["pubkey"])) != c["code"]]
    assert not bad, (l, bad[:5])
