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


PATHS = re.compile(r"^path counts: host (\d+) gpu (\d+) fallback (\d+)$", re.M)
PER_MODULE = re.compile(r"^(\w+): wall ([\d.]+) ms, drop-in calls: host (\d+), gpu (\d+)$", re.M)


@pytest.mark.gpu
@pytest.mark.parametrize("force", [None, "gpu"])
def test_reference_test_suite_passes_through_the_dropin(force):
    """bee2's acceptance suite (test/test.c:43-173 -> test/crypto/belt_test.c:178-215,423-472, bash_test.c:41-154, bign_test.c:338-400, ...)
    through the drop-in, twice: as a caller gets the library (KAT-sized calls take the host path by size), and under
    BEE2HIP_FORCE=gpu, where EVERY drop-in call must have gone to a kernel -- bee2hip_path_count(0) == 0 -- so the STB vectors are
    checked against the kernels themselves, not against host_small.hpp.  Per-module wall times are recorded."""
    env = dict(os.environ)
    env.pop("BEE2HIP_FORCE", None)
    if force:
        env["BEE2HIP_FORCE"] = force
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=1800, env=env)
    assert "bash_platform = BASH_HIP" in r.stdout, r.stdout + r.stderr
    for m in MODULES:
        assert f"{m}: OK" in r.stdout, r.stdout + r.stderr[-2000:]
    assert r.returncode == 0 and "Err" not in r.stdout
    host, gpu, fallback = (int(x) for x in PATHS.search(r.stdout).groups())
    per = {m.group(1): (float(m.group(2)), int(m.group(3)), int(m.group(4))) for m in PER_MODULE.finditer(r.stdout)}
    assert set(per) == set(MODULES) and fallback == 0
    if force == "gpu":
        assert host == 0 and gpu > 1000, (host, gpu)          # nothing took the host path; the suite makes thousands of drop-in calls
        for m in MODULES:
            assert per[m][1] == 0 and per[m][2] > 0, (m, per[m])
    else:
        assert host > 0                                       # the default: KAT-sized calls run on the calling core
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"r06_reftests_{force or 'auto'}.txt"), "w") as f:
        f.write(f"oracle/_ref/testbee2_hip (bee2's own test/crypto/*_test.c linked -lbee2hip first), BEE2HIP_FORCE={force or '(unset)'}\n" + r.stdout)


# bee2's bench lines the library is ALLOWED to lose against the reference on the same box (each is named in INTEGRATION.md
# "Known regressions of single calls" with its reason); everything else must be at least level (within SLACK, below).
# The test fails when a line NOT listed here is slower -- and also when a listed line stopped being slower, so that the
# list cannot outlive its reasons.
KNOWN_SLOWER = set()       # round 4: none (single signatures and key pairs moved to the constant-time host path)
# bee2's loops last tens of milliseconds: even the best of three runs scatters by +-10 % on a shared host, and several lines sit AT
# the reference by construction (the host path is a table-driven belt like bee2's own: belt-mac 0.96, belt-sde 0.98).  The record keeps
# the exact ratios; the test fails on a clear loss only.
SLACK = 0.8
BENCH_LINE = re.compile(r"^(\w+Bench::[\w-]+):\s+(\d+) ([\w/]+) \[\s*(\d+) ([\w/]+)\]", re.M)


def _bench(binp, env=None, runs=1):
    """-> (output of the last run, {line: (best rate over the runs, unit)}): bee2's loops are short (tens of milliseconds)
    and a single run scatters by +-20 % on a shared host, so each side gets the best of `runs`"""
    best, out = {}, ""
    for _ in range(runs):
        r = subprocess.run([binp, "bashbench", "beltbench", "bignbench"], capture_output=True, text=True, timeout=1800,
                           env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stdout + r.stderr[-2000:]
        out = r.stdout
        for m in BENCH_LINE.finditer(out):
            v = int(m.group(4))
            if v > best.get(m.group(1), (0, ""))[0]:
                best[m.group(1)] = (v, m.group(5))
    return out, best


@pytest.mark.gpu
def test_reference_bench_functions_run_through_the_dropin():
    """bashBench prints bash_platform (test/crypto/bash_bench.c:27,47 -- row a5) and times bee2's loops
    (test/crypto/belt_bench.c:83-127, bign_bench.c:79-158, bash_bench.c:47-107) over our symbols.  The same binary linked
    against the reference alone runs beside it on the same box; both outputs and the per-line ratio are RECORDED
    (gpurun_out/r04_reftests_bench.txt -> profiles/), in auto mode (what a caller gets) and under BEE2HIP_FORCE=gpu."""
    out_ref, ref = _bench(CTL, runs=3)
    out_auto, auto = _bench(BIN, runs=3)
    out_gpu, gpu = _bench(BIN, {"BEE2HIP_FORCE": "gpu"})
    assert "bashBench::platform = BASH_HIP" in out_auto
    for m in ("beltBench", "bashBench", "bignBench"):
        assert f"{m}: OK" in out_auto and f"{m}: OK" in out_gpu, out_auto + out_gpu
    lines = ["bee2's own bench functions (test/crypto/{belt,bash,bign}_bench.c, compiled where they lie) on this box:",
             "testbee2_ref = linked against the reference alone (one host core); testbee2_hip = linked -lbee2hip first.",
             "rate = the bracketed figure bee2 prints (kBytes/sec, sigs/sec, ...), best of 3 runs for the reference and for auto mode;",
             "ratio = hip / reference.", "",
             f"{'line':34s} {'reference':>12s} {'hip auto':>12s} {'ratio':>7s} {'hip FORCE=gpu':>14s} {'ratio':>7s}  unit"]
    slower, recovered = [], []
    for name, (r0, unit) in ref.items():
        a, g = auto.get(name, (0, unit))[0], gpu.get(name, (0, unit))[0]
        mark = ""
        if name.split("::")[1].startswith(("bash-prg", "belt-cfb", "KeyWrap", "KeyUnwrap")):
            mark = "  (not a drop-in symbol: bee2's own code on both sides)"
        elif a < SLACK * r0:
            mark = "  KNOWN regression (INTEGRATION.md)" if name in KNOWN_SLOWER else "  <-- SLOWER, not listed"
            if name not in KNOWN_SLOWER:
                slower.append((name, r0, a))
        elif name in KNOWN_SLOWER:
            recovered.append(name)
        lines.append(f"{name:34s} {r0:12d} {a:12d} {a / max(r0, 1):7.2f} {g:14d} {g / max(r0, 1):7.2f}  {unit}{mark}")
    lines += ["", "---- testbee2_ref", out_ref, "---- testbee2_hip (auto)", out_auto, "---- testbee2_hip (BEE2HIP_FORCE=gpu)", out_gpu]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_reftests_bench.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert not slower, f"slower than the reference on the same box and not a listed regression: {slower}"
    assert not recovered, f"listed as known regressions but no longer slower -- shrink KNOWN_SLOWER: {recovered}"
