import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def hostshim_san_flags():
    """extra g++ flags for the host-path shims (tests/hostshim/*.cpp = bee2_amd/csrc/host_*.hpp, product code): empty by default;
    tests/test_host_sanitizers.py re-runs the host-path modules with BEE2_HOSTSHIM_SAN=address,undefined (and the sanitizer
    runtime preloaded into the interpreter), the way the reference builds its ASan / check configurations
    (/root/reference/CMakeLists.txt:94-101,126-132)"""
    san = os.environ.get("BEE2_HOSTSHIM_SAN", "")
    return ["-g", "-fno-omit-frame-pointer", f"-fsanitize={san}", "-fno-sanitize-recover=all"] if san else []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference compiled in the build container)")


@pytest.fixture(scope="session")
def orc():
    import orclib
    return orclib.load()


@pytest.fixture(scope="session")
def golden():
    import orclib
    return orclib.Golden()


# ---- the drop-in layer has two ways to evaluate a small single call (bee2_amd/csrc/host_small.hpp): every -m gpu test of
# the modules below runs once with BEE2HIP_FORCE=gpu semantics (every primitive in a kernel: the coverage rounds 1-2 had)
# and once with =cpu (every drop-in call that has a host path takes it, at every size).  All other tests run in the
# default "auto" mode (crossover by size), which is what a caller gets.
DROPIN_TEST_MODULES = {"test_gpu_bash", "test_gpu_belt", "test_gpu_belt_bde", "test_gpu_belt_dwp", "test_gpu_belt_modes",
                       "test_gpu_belt_sde", "test_gpu_threads", "test_gpu_bign"}
FORCE_CODE = {"auto": 0, "gpu": 1, "cpu": 2}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__ in DROPIN_TEST_MODULES and "force_path" in metafunc.fixturenames:
        metafunc.parametrize("force_path", ["gpu", "cpu"], indirect=True)


@pytest.fixture(autouse=True)
def force_path(request):
    mode = getattr(request, "param", None)
    if mode is None or request.node.get_closest_marker("gpu") is None:
        yield "auto"
        return
    import bee2_amd
    lib = bee2_amd.load().lib
    lib.bee2hip_path_policy(FORCE_CODE[mode])              # product ABI (include/bee2hip.h): as BEE2HIP_FORCE
    before = [lib.bee2hip_path_count(i) for i in range(3)]
    yield mode
    after = [lib.bee2hip_path_count(i) for i in range(3)]
    lib.bee2hip_path_policy(0)
    # the mode did what it says: no host-path call under "gpu", no GPU drop-in helper call under "cpu", no fault fallback
    if mode == "gpu":
        assert after[0] == before[0], "host path taken under BEE2HIP_FORCE=gpu"
    else:
        assert after[1] == before[1], "GPU drop-in helper ran under BEE2HIP_FORCE=cpu"
    assert after[2] == before[2]
