"""bee2's OWN acceptance tests (test/crypto/{bash,belt,bign,bign128,bign192,bign256}_test.c of the reference,
compiled where they lie by `make -C oracle reftests`, never copied) running on the MI355X through the drop-in
symbols: oracle/_ref/testbee2_hip is linked -lbee2hip first, the compiled reference second, so every bee2
symbol the HIP library exports binds to it -- for the tests' own calls and for the calls the reference makes
internally.  SURVEY.md 8b; INTEGRATION.md section 2 acted out.  Test infrastructure only."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "testbee2_hip")
CTL = os.path.join(ROOT, "oracle", "_ref", "testbee2_ref")
MODULES = ("beltTest", "bashTest", "bignTest", "bign128Test", "bign192Test", "bign256Test")
# what must come from the HIP library when the reference's tests run (a subset of its exports: the hot path)
MUST_BIND = ("bashF", "bashHashStepH", "bashHash", "beltBlockEncr", "beltCTRStepE", "beltCTR", "beltMACStepA", "beltMAC",
             "beltHashStepH", "beltHash", "beltECBStepE", "beltCBCStepD", "beltBDEStepE", "beltSDEStepE", "beltDWPStepE",
             "beltDWPWrap", "beltCHEStepE", "beltCHEUnwrap", "bignVerify", "bignSign", "bignSign2", "bignKeypairGen",
             "bignPubkeyVal", "bignPubkeyCalc", "bign128Verify", "bign128Sign", "bign192Verify", "bign256Verify")

pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/testbee2_hip not built (make -C oracle reftests)")


def test_control_binary_passes_against_the_reference_alone():
    """the same objects linked against the reference only: no GPU needed, must print the same OK lines"""
    r = subprocess.run([CTL], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for m in MODULES:
        assert f"{m}: OK" in r.stdout


def test_reference_tests_bind_the_hot_path_to_the_hip_library():
    """LD_DEBUG=bindings with LD_BIND_NOW: which library each bee2 symbol resolves to (no call is made before main)"""
    env = dict(os.environ, LD_BIND_NOW="1", LD_DEBUG="bindings")
    r = subprocess.run([BIN, "none"], capture_output=True, text=True, timeout=300, env=env)
    bound = {}
    for m in re.finditer(r"binding file (\S+) \[\d+\] to (\S+) \[\d+\]: normal symbol `(\w+)'", r.stderr):
        src, dst, sym = m.groups()
        if os.path.basename(src) in ("testbee2_hip", "libbee2ref.so"):
            bound.setdefault(sym, set()).add(os.path.basename(dst))
    for s in MUST_BIND:
        assert bound.get(s) == {"libbee2hip.so"}, (s, bound.get(s))
    # and the reference's INTERNAL callers go to the GPU as well (e.g. beltWBL/KWP -> beltBlockEncr, brng -> beltHash)
    assert bound["beltBlockEncr2"] == {"libbee2hip.so"} and bound["beltKeyExpand2"] == {"libbee2hip.so"}


@pytest.mark.gpu
def test_reference_test_suite_passes_through_the_dropin():
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=1200)
    assert "bash_platform = BASH_HIP" in r.stdout, r.stdout + r.stderr
    for m in MODULES:
        assert f"{m}: OK" in r.stdout, r.stdout + r.stderr[-2000:]
    assert r.returncode == 0 and "Err" not in r.stdout


@pytest.mark.gpu
def test_reference_bench_functions_run_through_the_dropin():
    """bashBench prints bash_platform (test/crypto/bash_bench.c:27,47 -- row a5) and times bee2's loops over our symbols"""
    r = subprocess.run([BIN, "bashbench", "beltbench", "bignbench"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "bashBench::platform = BASH_HIP" in r.stdout
    for m in ("beltBench", "bashBench", "bignBench"):
        assert f"{m}: OK" in r.stdout, r.stdout
