"""CPU tests of bee2_amd/csrc/host_bign_ct.hpp -- the drop-in layer's CONSTANT-TIME host path for one public-key
calculation / key generation / signature (product code) -- against the reference's own outputs in the committed fixtures
(tests/golden/bign_sign.json: STB annex G vectors, keys 0 / 1 / q-1 / q / 2^2l-1, hashes at and beyond q, additional
input of 0..300 octets) and against the oracle on random items.  tests/hostshim/host_bign_ct_shim.cpp gives the header a
C view; this module builds it with g++ (no GPU, no HIP).  The same code is reached on the GPU box through bignSign2 /
bign128Sign2 / ... of libbee2hip.so in auto mode (tests/test_gpu_bign_sign.py runs its drop-in fixtures through both paths)."""
import ctypes
import json
import os
import random
import subprocess

import pytest

from conftest import hostshim_san_flags

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = {128: 189, 192: 317, 256: 569}
Q = {}


@pytest.fixture(scope="module")
def hc(tmp_path_factory, orc):
    out = tmp_path_factory.mktemp("hostshim") / "libhostbignct.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror"] + hostshim_san_flags() + ["-o", str(out),
                           os.path.join(ROOT, "tests", "hostshim", "host_bign_ct_shim.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.hc_pubkey_calc.restype = ctypes.c_uint32
    lib.hc_sign.restype = ctypes.c_uint32
    assert lib.hc_init(orc.beltH()) == 1
    return lib


@pytest.fixture(scope="module")
def fixtures():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bign_sign.json")))


def _sz(n):
    return ctypes.c_size_t(n)


def _field(hc, l, op, a, b=0):
    n = l // 4
    r = ctypes.create_string_buffer(n)
    hc.hc_field(_sz(l), op, r, a.to_bytes(n, "little"), b.to_bytes(n, "little"))
    return int.from_bytes(r.raw, "little")


@pytest.mark.parametrize("l", [128, 192, 256])
def test_constant_time_field_ops_against_python_integers(hc, l):
    """every operand is ANY 2l-bit number (values are kept weakly reduced); the corrections that are rare for random
    operands -- the second wrap of add / sub, the third fold of the reduction -- are forced with crafted ones"""
    p = 2 ** (2 * l) - C[l]
    top = 2 ** (2 * l)
    rnd = random.Random(l)
    corner = [0, 1, 2, C[l] - 1, C[l], C[l] + 1, p - 1, p, p + 1, top - 1, top - 2, top - C[l], top - C[l] - 1, 2 ** (2 * l - 1),
              2 ** 64 - 1, 2 ** 64, top - 2 ** 64, 2 ** 128 - 1, (top - 1) ^ (2 ** 64 - 1)]
    vals = corner + [rnd.randrange(top) for _ in range(120)]
    for a in vals:
        assert _field(hc, l, 5, a) == a % p                       # canon
        if a % p:
            assert _field(hc, l, 4, a) * a % p == 1
        for b in corner + [rnd.randrange(top) for _ in range(5)]:
            assert _field(hc, l, 0, a, b) == a * b % p
            assert _field(hc, l, 2, a, b) == (a + b) % p
            assert _field(hc, l, 3, a, b) == (a - b) % p
            # the raw results are proper N-limb numbers of the right class
            assert _field(hc, l, 6, a, b) % p == (a + b) % p and _field(hc, l, 7, a, b) % p == (a - b) % p
    assert _field(hc, l, 4, 0) == 0 and _field(hc, l, 4, p) == 0


@pytest.mark.parametrize("l", [128, 192, 256])
def test_mod_q_of_any_double_length_number(hc, l):
    q = _curve_q(l)
    n = l // 4
    rnd = random.Random(l + 1)
    top = 2 ** (4 * l)
    xs = [0, 1, q - 1, q, q + 1, 2 * q, top - 1, 2 ** (2 * l) - 1, 2 ** (2 * l), 2 ** (2 * l) + 1, (2 ** (2 * l) - 1) * q, (q - 1) ** 2,
          (2 ** (l + 1) - 1) * (q - 1)] + [rnd.randrange(top) for _ in range(300)] + [rnd.randrange(2 ** (3 * l + 1)) for _ in range(300)]
    for x in xs:
        r = ctypes.create_string_buffer(n)
        hc.hc_mod_q(_sz(l), r, x.to_bytes(2 * n, "little"))
        assert int.from_bytes(r.raw, "little") == x % q, hex(x)


def _curve_q(l):
    src = open(os.path.join(ROOT, "bee2_amd", "csrc", "bign_curves.inc")).read()
    import re
    m = re.search(r"k_bign%d_q\[[^\]]*\]\s*=\s*\{([^}]*)\}" % l, src)
    return int.from_bytes(bytes(int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", m.group(1))), "little")


@pytest.mark.parametrize("l", [128, 192, 256])
def test_pubkey_calc_and_keygen_on_the_reference_fixtures(hc, fixtures, l):
    no = l // 4
    for c in fixtures[str(l)]["pubkey_calc"]:
        pub = ctypes.create_string_buffer(b"\xEE" * (2 * no), 2 * no)
        code = hc.hc_pubkey_calc(_sz(l), 0, bytes.fromhex(c["priv"]), pub)
        assert code == c["code"], c["priv"]
        assert pub.raw.hex() == (c["pub"] if code == 0 else "ee" * (2 * no))      # a refused key writes nothing
    for c in fixtures[str(l)]["keypair_gen"]:
        if c["code"] == 0 and c.get("priv") and c["defined"]:      # (a draw in [q, p): the reference's own result is not d G)
            pub = ctypes.create_string_buffer(2 * no)
            assert hc.hc_pubkey_calc(_sz(l), 1, bytes.fromhex(c["priv"]), pub) == 0
            assert pub.raw.hex() == c["pub"]
    # key generation multiplies ANY d (the reference draws below p, not q): d = q gives the point at infinity -> ERR_BAD_PARAMS
    q = _curve_q(l)
    pub = ctypes.create_string_buffer(2 * no)
    assert hc.hc_pubkey_calc(_sz(l), 1, q.to_bytes(no, "little"), pub) == 502
    assert hc.hc_pubkey_calc(_sz(l), 1, (0).to_bytes(no, "little"), pub) == 502
    assert hc.hc_pubkey_calc(_sz(l), 0, q.to_bytes(no, "little"), pub) == 504


@pytest.mark.parametrize("l", [128, 192, 256])
def test_sign2_on_the_reference_fixtures(hc, fixtures, l):
    no, sg = l // 4, 3 * l // 8
    ran = 0
    for c in fixtures[str(l)]["sign2"]:
        oid = bytes.fromhex(c["oid"])
        if c["code"] not in (0, 504):
            continue                                  # OID / pointer errors are the C ABI's business (capi.hip), not this header's
        t = bytes.fromhex(c["t"]) if c["t"] is not None else None
        sig = ctypes.create_string_buffer(sg)
        code = hc.hc_sign(_sz(l), oid, _sz(len(oid)), bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]), None, t,
                          _sz(len(t) if t else 0), sig)
        assert code == c["code"], c
        if code == 0:
            assert sig.raw.hex() == c["sig"], c
        ran += 1
    assert ran >= 20


@pytest.mark.parametrize("l", [128, 192, 256])
def test_sign_with_given_one_time_keys_against_the_oracle(hc, orc, l):
    """bignSign after its generator: k supplied.  Random (d, k, H) plus the corners k = 1, q - 1, d = 1, q - 1, H >= q"""
    from bee2_amd import engine as E
    no, sg = l // 4, 3 * l // 8
    q = _curve_q(l)
    oid = E.LEVEL_OID[l]
    rnd = random.Random(l + 7)
    cases = [(rnd.randrange(1, q), rnd.randrange(1, q), rnd.randrange(2 ** (2 * l))) for _ in range(40)]
    cases += [(1, 1, 0), (q - 1, q - 1, 2 ** (2 * l) - 1), (q - 1, 1, q), (1, q - 1, q + 12345), (q - 2, 2, q - 1), (2, q - 2, 1)]
    for d, k, h in cases:
        sig = ctypes.create_string_buffer(sg)
        db, kb, hb = d.to_bytes(no, "little"), k.to_bytes(no, "little"), h.to_bytes(no, "little")
        assert hc.hc_sign(_sz(l), oid, _sz(len(oid)), hb, db, kb, None, _sz(0), sig) == 0
        code, want, used = orc.sign_rnd(l, oid, hb, db, kb)           # the generator's first draw is k
        assert code == 0 and used == 1 and sig.raw == want, (d, k, h)
    # a one-time key outside (0, q) is the generator's fault
    sig = ctypes.create_string_buffer(sg)
    one = (1).to_bytes(no, "little")
    assert hc.hc_sign(_sz(l), oid, _sz(len(oid)), one, one, (0).to_bytes(no, "little"), None, _sz(0), sig) == 304
    assert hc.hc_sign(_sz(l), oid, _sz(len(oid)), one, one, q.to_bytes(no, "little"), None, _sz(0), sig) == 304
    assert hc.hc_sign(_sz(l), oid, _sz(len(oid)), one, q.to_bytes(no, "little"), one, None, _sz(0), sig) == 504


@pytest.mark.parametrize("l", [128, 192, 256])
def test_sign2_random_against_the_oracle(hc, orc, l):
    from bee2_amd import engine as E
    no, sg = l // 4, 3 * l // 8
    q = _curve_q(l)
    oid = E.LEVEL_OID[l]
    rnd = random.Random(l + 9)
    for i in range(60):
        d = rnd.randrange(1, q).to_bytes(no, "little")
        h = rnd.randbytes(no)
        # (256 octets: the oracle's own limit; 300-octet inputs are among the reference fixtures above)
        t = None if i % 3 == 0 else rnd.randbytes(rnd.choice((1, 31, 32, 33, 64, 65, 256)))
        sig = ctypes.create_string_buffer(sg)
        assert hc.hc_sign(_sz(l), oid, _sz(len(oid)), h, d, None, t, _sz(len(t) if t else 0), sig) == 0
        assert (0, sig.raw) == orc.sign2(l, oid, h, d, t), i


def test_compiled_arithmetic_has_no_branch_on_data():
    """tools/ct_audit_x86.py, machine-checked part: in the object the PRODUCT's host compiler makes (hipcc's clang, -O3) the
    functions that see secrets -- FieldCt add / sub / mul / sqr / reduce / canon / inv, jac_madd, mul_base, mod_q, sub_mod_q --
    contain nothing but loop back-edges and the windows of the public exponent p - 2: no forward conditional jump, no jump on
    the carry flag.  (clang -O3 had turned `x + (c & -carry)` into `if (carry) x += c` until the masks were hidden from it:
    host_bign_ct.hpp m_hide.)  With g++ -O2 -- what the tests above ran -- no jump on the carry / overflow flag."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ct_audit_x86", os.path.join(ROOT, "tools", "ct_audit_x86.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    bad, seen = audit.violations(0)
    if bad is None:
        pytest.skip("the ROCm clang is not installed here")
    assert seen >= 18 and not bad, "\n".join(bad)
    bad, seen = audit.violations(1, carry_only=True)
    assert bad is not None and seen >= 12 and not bad, "\n".join(bad)
