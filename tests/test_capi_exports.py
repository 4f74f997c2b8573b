"""CPU tests of the drop-in boundary: libbee2hip.so builds for gfx950 without a GPU,
loads, and exports every symbol include/bee2hip.h declares.  No compute calls here."""
import ctypes
import os
import re
import shutil

import pytest

import bee2_amd
from bee2_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(E.LIB_PATH):
        if shutil.which("hipcc") is None:
            pytest.skip("libbee2hip.so not built and hipcc unavailable")
        E.build()
    return E.LIB_PATH


def header_symbols(name="bee2hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))
    names |= set(re.findall(r"extern const char (\w+)\[\]", text))
    keep = {n for n in names if n.startswith(("bee2hip_", "bash", "belt", "bign"))}
    keep.discard("beltCTRStepD")           # a macro alias, as in belt.h:724
    return keep


def test_header_lists_the_expected_interface():
    syms = header_symbols()
    assert set(E.DROPIN_SYMBOLS) <= syms
    assert set(E.BATCH_SYMBOLS) <= syms


def test_library_exports_every_declared_symbol(lib_path):
    exported = E.lib_exports(lib_path)
    missing = sorted(header_symbols() - exported)
    assert not missing, f"declared in include/bee2hip.h but not exported: {missing}"


def test_product_library_ships_no_hooks_and_no_experiment_kernels(lib_path):
    """the library a bee2 maintainer links exports nothing of include/bee2hip_internal.h, and its code
    object holds only kernels a product entry point dispatches (no rejected A/B variants, no debug / probe kernels)"""
    exported = E.lib_exports(lib_path)
    hooks = sorted(n for n in exported if "internal" in n or "debug" in n or "time_kernel" in n or "clock_probe" in n)
    assert not hooks, hooks
    assert not (header_symbols("bee2hip_internal.h") & exported)
    import subprocess
    names = subprocess.check_output([os.path.join(ROOT, "tools", "list_kernels.sh"), lib_path], text=True).splitlines()
    kernels = sorted({n.split("(")[0] for n in names if n.strip()})
    assert kernels and len(kernels) <= 148, len(kernels)          # 124 + the one-signer / few-signers family (key table, main in three forms x 3 curves, the fallback's key replication)
    for bad in ("debug", "clock_probe", "bashF_batch_kernel", "bashF_walk_kernel", "BeltTabHyb", "BeltTabTwoS", "BeltTabTwoL",
                "BeltTabTwoQ"):
        assert not [k for k in kernels if bad in k], bad
    assert len([k for k in kernels if "bashF_tile_kernel" in k]) == 1
    assert len([k for k in kernels if "beltCTR_blocks_kernel" in k]) == 2          # with / without the hoisted round-1 G-box
    assert len([k for k in kernels if "bign_mulbase_ct_kernel" in k]) == 3         # one per curve


def test_experiments_library_exports_the_hooks():
    if not os.path.exists(E.EXP_LIB_PATH):
        pytest.skip("libbee2hip_exp.so not built")
    exported = E.lib_exports(E.EXP_LIB_PATH)
    assert not sorted(header_symbols() - exported)
    assert not sorted(header_symbols("bee2hip_internal.h") - exported)


def test_product_header_has_no_test_hooks():
    """the product ABI is bee2's names + the batch API; timing / debug / tuning hooks live in
    include/bee2hip_internal.h"""
    prod = header_symbols()
    internal = header_symbols("bee2hip_internal.h")
    assert internal == set(E.INTERNAL_SYMBOLS)
    assert not (prod & internal)
    assert not [n for n in prod if "debug" in n or "internal" in n or "time_kernel" in n]


def test_library_loads_and_reports_version(lib_path):
    lib = ctypes.CDLL(lib_path)
    lib.bee2hip_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.bee2hip_version()
    lib.bashF_deep.restype = ctypes.c_size_t
    assert lib.bashF_deep() == 0                           # bash_f64.c:189-192 contract
    name = (ctypes.c_char * 16).in_dll(lib, "bash_platform").value
    assert name.startswith(b"BASH_HIP")
    # state sizes equal bee2's (bash_hash.c:25-36, belt_lcl.h:135-141, belt_mac.c:32-45)
    for fn, want in (("bashHash_keep", 400), ("beltCTR_keep", 72), ("beltMAC_keep", 104)):
        f = getattr(lib, fn)
        f.restype = ctypes.c_size_t
        assert f() == want, fn


def test_sbox_generated_by_the_library_matches_standard(lib_path, golden):
    lib = ctypes.CDLL(lib_path)
    lib.beltH.restype = ctypes.POINTER(ctypes.c_ubyte)
    p = lib.beltH()
    assert bytes(p[i] for i in range(256)) == golden.H


def test_argument_checks_need_no_gpu(lib_path):
    """error paths that bee2 takes before touching data (bash_hash.c:122-123,
    belt_ctr.c:117-123, bign_params.c:244-280, oid.c:94-101)"""
    eng = bee2_amd.load(lib_path)
    assert eng.bashHash(0, b"")[0] == E.ERR_BAD_PARAMS
    assert eng.bashHash(100, b"")[0] == E.ERR_BAD_PARAMS
    assert eng.beltCTR(b"x" * 16, b"k" * 17, b"i" * 16)[0] == E.ERR_BAD_INPUT
    assert eng.beltMAC(b"x", b"k" * 5)[0] == E.ERR_BAD_INPUT
    p = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    assert p.l == 128 and bytes(p.p)[:2] == b"\x43\xff"
    with pytest.raises(E.EngineError):
        eng.bignParamsStd("1.2.3")
    h, s, k = b"\0" * 32, b"\0" * 48, b"\0" * 64
    assert eng.bignVerify(p, b"\x06\x01", h, s, k) == E.ERR_BAD_OID          # truncated DER
    assert eng.bignVerify(p, b"\x05\x00", h, s, k) == E.ERR_BAD_OID          # wrong tag
    assert eng.bignVerify(p, b"\x06\x02\x80\x01", h, s, k) == E.ERR_BAD_OID  # leading 0x80
    bad = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    bad.p[0] = 0x42
    assert eng.bignVerify(bad, E.OID_BELT_HASH_DER, h, s, k) == E.ERR_BAD_PARAMS
    other = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    other.b[0] ^= 1                       # a different (valid-looking) curve: served by the general-curve kernels, so
    # without a GPU the call must fail loudly with the device error (there is no CPU fallback), with one it verifies
    assert eng.bignVerify(other, E.OID_BELT_HASH_DER, h, s, k) in (E.ERR_BEE2HIP_DEVICE, E.ERR_BAD_SIG)
    assert eng.bignVerify(other, b"\x06\x01", h, s, k) == E.ERR_BAD_OID         # still before any device work
    other.a[0:32] = other.p[0:32]         # a = p: what bignEcCreate rejects (bign_ec.c:64-70), before inputs and OID
    assert eng.bignVerify(other, b"\x06\x01", h, s, k) == E.ERR_BAD_PARAMS
    assert eng.bignPubkeyVal(other, None if False else k) == E.ERR_BAD_PARAMS


def test_oid_der_validation_matches_reference_on_invalid_inputs(lib_path):
    """every DER string the reference's oidFromDER rejects must give ERR_BAD_OID before any
    device work (valid ones are exercised on the GPU in tests/test_gpu_bign.py)"""
    import json
    eng = bee2_amd.load(lib_path)
    p = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "oid_der_cases.json")))
    h, s, k = b"\0" * 32, b"\0" * 48, b"\0" * 64
    bad = [c for c in cases if not c["valid"]]
    assert len(bad) > 100
    for c in bad:
        der = bytes.fromhex(c["der"])
        assert eng.bignVerify(p, der, h, s, k) == E.ERR_BAD_OID, c["der"]


# entries whose whole body cannot throw (a constant, a sizeof, a thread-local pointer, two atomic loads): the only ones allowed
# without a function-try-block
NO_THROW_BY_INSPECTION = {"bee2hip_last_error", "bee2hip_version", "bashF_deep", "bashHash_keep", "beltCTR_keep", "beltMAC_keep",
                          "beltECB_keep", "beltDWP_keep", "beltHash_keep", "beltSDE_keep", "beltCHE_keep", "beltBDE_keep",
                          "beltCBC_keep", "bee2hip_path_policy", "bee2hip_path_count", "bee2hip_internal_stat", "bee2hip_device_count"}


def test_nothing_unwinds_through_the_c_abi_by_construction():
    """every extern "C" entry point is a function-try-block that ends in one of staging.hpp's B2H_CATCH
    macros (bad_alloc -> ERR_OUTOFMEMORY, anything else -> the device error code), read off the sources -- a new entry
    without one fails here, on CPU.  (tests/test_gpu_oom_injection.py makes the n-th allocation fail on the GPU box.)"""
    import glob
    src = sorted(glob.glob(os.path.join(ROOT, "bee2_amd", "csrc", "capi*.hip"))) + [os.path.join(ROOT, "bee2_amd", "csrc", "multi.hip")]
    assert len(src) >= 7
    entries, guarded, bare = set(), set(), set()
    for path in src:
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            if not line.startswith('extern "C"') or "(" not in line or "[] =" in line or line.rstrip().endswith(";"):
                continue
            name = re.match(r'extern "C" .+?\b([A-Za-z_][A-Za-z0-9_]*)\(', line).group(1)
            entries.add(name)
            text = "\n".join(lines[i: i + 12])
            pos, depth = text.index("(", text.index(name)), 0
            while True:                                        # the parenthesis that closes the parameter list
                depth += {"(": 1, ")": -1}.get(text[pos], 0)
                pos += 1
                if depth == 0:
                    break
            if re.match(r"\s*try \{", text[pos:]):
                rest = "\n".join(lines[i:])
                m = re.search(r"^\} (B2H_CATCH|catch \(\.\.\.\))|\} (B2H_CATCH(_VOID|_FALSE)?\b|catch \(\.\.\.\))[^\n]*$", rest, flags=re.M)
                nxt = rest.find('\nextern "C"', 1)
                assert m and (nxt < 0 or m.start() < nxt), (path, name)      # the catch belongs to THIS entry
                guarded.add(name)
            else:
                bare.add(name)
    # the level-fixed facades of capi_bign.hip come out of one macro: bign<L>PubkeyCalc / KeypairGen / Sign / Sign2 for L = 128, 192, 256
    for path in src:
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            m = re.match(r'\s+extern "C" err_t bign##L##(\w+)\(', line)
            if m:
                assert re.match(r"\s+try \{.*\} B2H_CATCH\b", lines[i + 1]), (path, m.group(1))
                for lvl in (128, 192, 256):
                    entries.add(f"bign{lvl}{m.group(1)}")
                    guarded.add(f"bign{lvl}{m.group(1)}")
    assert bare <= NO_THROW_BY_INSPECTION, sorted(bare - NO_THROW_BY_INSPECTION)
    assert len(guarded) >= 130 and guarded | bare == entries
    # and every symbol the headers declare is one of them
    declared = header_symbols() | header_symbols("bee2hip_internal.h")
    declared.discard("bash_platform")
    assert declared <= entries, sorted(declared - entries)


def test_dynamic_symbols_are_the_header_and_nothing_else(lib_path):
    """hipcc gives every kernel's host stub default visibility, so 146 mangled _ZN7bee2hip... symbols used
    to be exported beside the C ABI; bee2_amd/csrc/exports.map makes them local.  What is left DEFINED in the dynamic table is
    exactly what include/bee2hip.h declares (the rest of `nm -D` are the undefined imports: HIP runtime, libc, libstdc++)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    defined = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert not [s for s in defined if s.startswith("_Z")], "mangled symbols exported"
    assert defined == header_symbols(), sorted(defined ^ header_symbols())
    if os.path.exists(E.EXP_LIB_PATH):
        out = subprocess.check_output(["nm", "-D", "--defined-only", E.EXP_LIB_PATH], text=True)
        d2 = {l.split()[-1] for l in out.splitlines() if l.strip()}
        assert d2 == header_symbols() | header_symbols("bee2hip_internal.h"), sorted(d2 ^ (header_symbols() | header_symbols("bee2hip_internal.h")))
