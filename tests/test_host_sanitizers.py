"""Sanitizer builds of the HOST-side product code (the reference's own ASan / "check" configurations,
/root/reference/CMakeLists.txt:94-101,126-132, .github/workflows/build.yml:14-38).  No GPU:

* AddressSanitizer + UndefinedBehaviorSanitizer over the host path of small single calls -- bee2_amd/csrc/host_small.hpp,
  host_bign.hpp, host_bign_ct.hpp through tests/hostshim/*_shim.cpp -- by re-running tests/test_host_{small,bign,bign_ct}.py in a
  child interpreter with the shims built -fsanitize=address,undefined and the runtimes preloaded;
* ThreadSanitizer and ASan/UBSan over staging.hpp (scratch pool, host fallback, the duplex pipeline's two threads, streams and
  events) and multi.hip (the persistent worker pool) compiled against tests/hostshim/mockhip -- a CPU stand-in for the HIP
  runtime whose streams are real threads -- driven by tests/hostshim/staging_mock_main.cpp;
* ThreadSanitizer over the oracle's persistent thread pool (oracle/orc_threads.c, the all-cores CPU baseline of bench.py).
A canary proves each harness really reports: a deliberate overflow / race must make it fail."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "hostshim")


def _rt(name):
    p = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


ASAN, UBSAN, TSAN = _rt("libasan.so"), _rt("libubsan.so"), _rt("libtsan.so")
CXX = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


@pytest.mark.skipif(not (ASAN and UBSAN), reason="gcc's libasan / libubsan not installed")
def test_host_path_modules_pass_under_asan_and_ubsan(tmp_path):
    env = dict(os.environ, BEE2_HOSTSHIM_SAN="address,undefined", LD_PRELOAD=f"{ASAN}:{UBSAN}",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the canary first: a shim-style library built with the same flags must be caught reading past a heap block
    src = tmp_path / "canary.cpp"
    src.write_text('#include <stdlib.h>\nextern "C" int canary(int i) { volatile char *p = (char *)malloc(8); int v = p[8 + i]; free((void *)p); return v; }\n')
    so = tmp_path / "libcanary.so"
    subprocess.check_call(["g++", "-shared", "-fPIC", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(so), str(src)])
    r = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({str(so)!r}).canary(0)"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr, r.stderr[-1500:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_host_small.py"), os.path.join(ROOT, "tests", "test_host_bign.py"),
                        os.path.join(ROOT, "tests", "test_host_bign_ct.py")], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def _build_mock(tmp_path, san, extra=()):
    out = tmp_path / f"staging_{san.replace(',', '_')}"
    subprocess.check_call(CXX + [f"-fsanitize={san}", "-fno-sanitize-recover=all", "-I", os.path.join(SHIM, "mockhip"), "-x", "c++"] + list(extra)
                          + ["-o", str(out), os.path.join(SHIM, "staging_mock_main.cpp")])
    return str(out)


@pytest.mark.skipif(not TSAN, reason="gcc's libtsan not installed")
def test_staging_and_worker_pool_are_race_free_under_tsan(tmp_path):
    """duplex_inplace (caller thread + helper thread + two streams + 2 x chunks events), scratch_for_stream from eight threads,
    with_host's retry / fallback, run_on_devices from four host threads: ThreadSanitizer must report nothing."""
    # canary: the harness sees a race when there is one (two threads, one plain int)
    src = tmp_path / "race.cpp"
    src.write_text("#include <thread>\nint x; int main() { std::thread a([] { for (int i = 0; i < 100000; ++i) x++; }); for (int i = 0; i < 100000; ++i) x++; a.join(); return 0; }\n")
    subprocess.check_call(["g++", "-O0", "-g", "-pthread", "-fsanitize=thread", "-o", str(tmp_path / "race"), str(src)])
    r = subprocess.run([str(tmp_path / "race")], capture_output=True, text=True, timeout=120)
    assert "ThreadSanitizer: data race" in r.stderr
    exe = _build_mock(tmp_path, "thread")
    for _ in range(3):                                   # schedules differ from run to run
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
        assert r.returncode == 0 and "staging mock ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]


@pytest.mark.skipif(not (ASAN and UBSAN), reason="gcc's libasan / libubsan not installed")
def test_staging_and_worker_pool_under_asan_and_ubsan(tmp_path):
    """the same driver with "device" memory on the heap: an overrun of a staging block, a use of a freed scratch block or an event
    destroyed under a queued task would be an ASan report"""
    exe = _build_mock(tmp_path, "address,undefined")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "staging mock ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.skipif(not TSAN, reason="gcc's libtsan not installed")
def test_oracle_thread_pool_is_race_free_under_tsan(tmp_path):
    srcs = [os.path.join(ROOT, "oracle", f) for f in sorted(os.listdir(os.path.join(ROOT, "oracle"))) if f.endswith(".c") and f != "ref_tests_main.c"]
    exe = tmp_path / "tsan_pool"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-pthread", "-fsanitize=thread", "-o", str(exe), os.path.join(SHIM, "tsan_pool_main.c")] + srcs)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tsan pool ok" in r.stdout, r.stdout + r.stderr[-4000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
