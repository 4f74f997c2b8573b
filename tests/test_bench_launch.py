"""`python bench.py --gpus N` must fan out to N ranks by itself (how the driver calls it): the launcher
re-executes the command line under torch.distributed.run on 127.0.0.1.  Run here on CPU with the gloo
backend and --launch-selftest (process group + one all-reduce, no GPU work)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BEE2_BENCH_BACKEND="gloo", **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_self_launches_n_ranks(n):
    r = _run(["--gpus", str(n), "--launch-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["roofline"]["n_ranks_seen"] == n and d["max_rank"] == n - 1
    assert d["roofline"]["n_devices_distinct"] == n          # (each rank reports its own stand-in device on CPU)
    assert d["only"] == "bashF,ctr,verify,mixed"             # N > 1 default: the four BASELINE workloads, not the N = 1 legs
    # the strong (fixed-N) split of SURVEY 8e: the ranks' shard_range shares of each BASELINE job add up to the job, and the line
    # of an N-rank run carries one measured speedup per workload
    assert d["strong_keys"] == ["strong_speedup_bashF", "strong_speedup_ctr", "strong_speedup_verify", "strong_speedup_mixed"]
    assert d["roofline"]["strong_items_bashF"] == 1 << 20 and d["roofline"]["strong_items_ctr"] == 1 << 30
    assert d["roofline"]["strong_items_verify"] == 1 << 18 and d["roofline"]["strong_items_mixed"] == 1 << 24
    assert "torch.distributed.run" in r.stderr and f"--nproc-per-node={n}" in r.stderr


def test_two_ranks_on_one_device_are_counted_as_one_and_refused_under_rccl():
    """ranks all-gather the identity of the device they drive.  With a mocked device list 0,0 the
    line must say n_devices_distinct 1 under gloo (the documented one-GPU rehearsal) and the run must REFUSE under the
    RCCL rule (one rank per GPU), instead of reporting n_gpus 2 from one card."""
    r = _run(["--gpus", "2", "--launch-selftest"], {"BEE2_BENCH_MOCK_DEVICES": "0,0"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["roofline"]["n_ranks_seen"] == 2 and d["roofline"]["n_devices_distinct"] == 1
    r = _run(["--gpus", "2", "--launch-selftest"], {"BEE2_BENCH_MOCK_DEVICES": "0,0", "BEE2_BENCH_MOCK_BACKEND": "nccl"})
    assert r.returncode != 0 and "2 ranks on 1 distinct device" in r.stderr
    r = _run(["--gpus", "3", "--launch-selftest"], {"BEE2_BENCH_MOCK_DEVICES": "0,1,2", "BEE2_BENCH_MOCK_BACKEND": "nccl"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["roofline"]["n_devices_distinct"] == 3


def test_distinct_device_rule_and_mixed_roofline_arithmetic():
    """pure functions of bench.py: the device rule, and configs[4]'s roofline from its two parts"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.check_distinct_devices(["a", "b", "c"], 3, "nccl") == 3
    assert b.check_distinct_devices(["a", "a"], 2, "gloo") == 1
    with pytest.raises(SystemExit):
        b.check_distinct_devices(["a", "a"], 2, "nccl")
    # a card's identity = its index AND what the runtime reports: cards that all report one uuid still count by index, ranks
    # that each see their card as index 0 are told apart by uuid, two ranks on one card are one device
    class P:
        def __init__(self, uuid, bus): self.uuid, self.pci_bus_id = uuid, bus
    props = {}
    real = b.torch.cuda.get_device_properties
    b.torch.cuda.get_device_properties = lambda i: props[i]
    try:
        props.update({0: P("00000000", 1), 1: P("00000000", 1)})
        assert b.check_distinct_devices([b.device_identity(0), b.device_identity(1)], 2, "nccl") == 2
        ids = []
        for u in ("GPU-aa", "GPU-bb"):
            props[0] = P(u, 3)
            ids.append(b.device_identity(0))
        assert b.check_distinct_devices(ids, 2, "nccl") == 2
        props[0] = P("GPU-aa", 3)
        assert b.check_distinct_devices([b.device_identity(0), b.device_identity(0)], 2, "gloo") == 1
    finally:
        b.torch.cuda.get_device_properties = real
    # round 3's numbers: 11.05 G perm/s and 941 GiB/s of blocks -> sum of parts 100.4 M msg/s, overlap ceiling 170 M
    r = b.mixed_roofline(99.9e6, 11.05e9, 941.4 * 2 ** 30 / 16, "test")
    assert abs(r["sum_of_parts_ceiling"] / 1e6 - 100.5) < 0.5 and abs(r["peak"] / 1e6 - 170.0) < 0.5
    assert 0.58 < r["frac"] < 0.60 and 0.97 < r["frac_sum_of_parts"] < 1.0 and -0.05 < r["overlap_got"] < 0.05
    # a fused kernel as fast as its slower part alone has all of the overlap
    r = b.mixed_roofline(170.0e6, 11.05e9, 941.4 * 2 ** 30 / 16, "test")
    assert abs(r["frac"] - 1.0) < 1e-3 and abs(r["overlap_got"] - 1.0) < 1e-2
    v = b.valu_picture(11.0e9, b.BASHF_VALU, 1.767)
    assert v["clock_measured"] and 1.0 < v["model_ratio"] < 1.3          # the model is not a ceiling at the real clock


def test_bench_refuses_a_world_size_that_is_not_gpus():
    # a launcher that started 2 ranks for --gpus 3 must not get n_gpus: 3 (nor a silent n_gpus: 2)
    env = dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    r = _run(["--gpus", "3", "--launch-selftest"], env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_refuses_more_gpus_than_the_node_has():
    # RCCL backend (the default): --gpus 9 on a node without nine devices is an error, not a smaller run
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "BEE2_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "9"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode != 0 and "refusing" in r.stderr


# what the driver's record must keep: `roofline` = 24 flat scalars in a fixed order (bench_legs/line.py) -- the headline kernel's
# figures, the three other BASELINE rates with their fractions, the 8-way strong split (predicted at N = 1, measured at N > 1),
# ranks / devices seen -- inside ONE strict-JSON line of at most 8000 characters (the record keeps 8081 characters of stdout)
ROOFLINE_24_N1 = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                  "beltCTR_GiBps", "beltCTR_frac", "bignVerify_sigs_per_s", "bignVerify_frac", "mixed_msgs_per_s", "mixed_frac",
                  "strong_pred_8_bashF", "strong_pred_8_ctr", "strong_pred_8_verify", "strong_pred_8_mixed",
                  "n_ranks_seen", "n_devices_distinct", "valu_busy", "frac_2p22", "beltCTR_lds_frac", "weak_efficiency")
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
LINE_MAX = 8000


def _strict(s):
    """json.loads that refuses NaN / Infinity (json.dumps' default would print them; a strict parser then drops the line)"""
    def no(c):
        raise ValueError(f"non-strict JSON constant {c}")
    return json.loads(s, parse_constant=no)


def _last_line(stdout):
    lines = stdout.rstrip("\n").split("\n")
    return lines[-1]


def _check_line_shape(raw, n):
    assert len(raw) <= LINE_MAX, len(raw)
    d = _strict(raw)
    for k in CONTRACT_KEYS:
        assert k in d, k
    want = tuple(k.replace("strong_pred_8_", "strong_speedup_") for k in ROOFLINE_24_N1) if n > 1 else ROOFLINE_24_N1
    assert tuple(list(d["roofline"])[:24]) == want, list(d["roofline"])[:24]
    for k, v in d["roofline"].items():                 # flat: no nested object, no prose beyond the kernel's name
        assert not isinstance(v, (dict, list)), k
        assert not isinstance(v, str) or len(v) <= 60, k
    for k, v in d["cpu_baseline"].items():
        assert not isinstance(v, (dict, list)), k
        assert not isinstance(v, str) or len(v) <= 120, k
    assert isinstance(d["config"].get("workload"), str) and "model" not in d["config"]
    return d


@pytest.mark.parametrize("n", [1, 8])
def test_line_is_strict_json_under_8000_characters_with_the_contract_keys(n):
    """the LAST stdout line of an N-rank run, built by the very code the real run uses
    (bench_legs/line.py) from stand-in numbers of the real magnitudes, parses strictly, is short enough for the driver's record and
    carries the contract's keys with `roofline`'s 24 scalars in order.  N = 8 is the rehearsal of the first real 8-GPU launch: eight
    gloo ranks, a mocked list of eight distinct devices under RCCL's one-rank-per-GPU rule, the shares of the four fixed jobs adding up."""
    env = {"BEE2_BENCH_MOCK_DEVICES": ",".join(str(i) for i in range(n)), "BEE2_BENCH_MOCK_BACKEND": "nccl"} if n > 1 else {}
    r = _run(["--gpus", str(n), "--launch-selftest"], env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = _last_line(r.stdout)
    d = _check_line_shape(raw, n)
    assert d["n_gpus"] == n and d["roofline"]["n_ranks_seen"] == n and d["roofline"]["n_devices_distinct"] == n
    assert d["roofline"]["strong_items_bashF"] == 1 << 20 and d["roofline"]["strong_items_ctr"] == 1 << 30
    assert d["roofline"]["strong_items_verify"] == 1 << 18 and d["roofline"]["strong_items_mixed"] == 1 << 24
    if n > 1:
        assert d["cpu_baseline"]["value"] is None and "N=1 only" in d["cpu_baseline"]["sample"]
        assert set(d["ranks"]) >= {"solo_value", "weak_efficiency", "clock_ghz_min", "clock_ghz_max"}
        # eight ranks on seven cards: refused under RCCL's rule
        r = _run(["--gpus", "8", "--launch-selftest"], {"BEE2_BENCH_MOCK_DEVICES": "0,1,2,3,4,5,6,6", "BEE2_BENCH_MOCK_BACKEND": "nccl"}, timeout=600)
        assert r.returncode != 0 and "8 ranks on 7 distinct device" in r.stderr


def test_line_builder_refuses_non_finite_numbers_and_oversize_lines():
    from bench_legs import line as L
    res = {"metric": "m", "value": float("nan"), "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": float("inf"),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": {"workload": "w"},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 2.0, "unit": "GB/s", "frac": 0.5, "traffic": None, "kernel": "k" * 500}}
    ln, detail = L.build(res, {}, {}, {}, 1, 1, 1, 1000.0)
    s = L.dumps(ln)
    d = _strict(s)
    assert d["value"] is None and d["ms_per_step"] is None and len(d["roofline"]["kernel"]) <= 60
    assert tuple(d["roofline"]) == ROOFLINE_24_N1
    json.dumps(detail, allow_nan=False)
    ln["config"]["workload"] = "x" * 9000
    with pytest.raises(AssertionError):
        L.dumps(ln)
    # numbers keep 8 significant digits, integers stay exact
    assert L.clean(402653184) == 402653184 and L.clean(0.123456789123) == 0.12345679


@pytest.mark.gpu
def test_bench_line_on_the_gpu_is_short_strict_and_complete():
    """the real thing, short: the four BASELINE legs on 1 GiB of stream, 3 steps; the LAST stdout line is the contract's line (<= 8000
    characters, strict JSON), the figures are in range, and the full record is in gpurun_out/bench_detail.json"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "BEE2_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--ctr-gib", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _check_line_shape(_last_line(r.stdout), 1)
    assert d["metric"] == "bashF perms/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    rf = d["roofline"]
    for k in ROOFLINE_24_N1:
        if k not in ("bound", "unit", "kernel", "traffic", "valu_busy", "weak_efficiency"):
            assert isinstance(rf[k], (int, float)), k
    assert rf["weak_efficiency"] is None                   # N = 1: nothing to compare with
    # `traffic` is MEASURED IN THIS RUN when rocprofv3 is on the box (bench_legs/pmc_live.py re-runs the headline leg under --pmc:
    # FETCH_SIZE and WRITE_SIZE passes, FETCH doubled): within a few per cent of the 384 B x 2^20 algorithmic bytes
    import shutil
    if shutil.which("rocprofv3"):
        assert 0.95 * 384 * (1 << 20) < rf["traffic"] < 1.10 * 384 * (1 << 20), rf["traffic"]
    else:
        assert rf["traffic"] is None or rf["traffic"] > 0  # (a replay of the committed profile of the same launch, or null)
    for k in ("frac", "beltCTR_frac", "beltCTR_lds_frac", "bignVerify_frac", "mixed_frac", "frac_2p22"):
        assert 0 < rf[k] < 1.05, (k, rf[k])
    for w in ("bashF", "ctr", "verify", "mixed"):          # a split can never be predicted to beat 8x by much (a share can run at a better clock)
        assert 1.0 < rf[f"strong_pred_8_{w}"] <= 9.0, w
    assert abs(rf["achieved"] - 384 * (1 << 20) / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-3 * rf["achieved"]
    assert rf["n_ranks_seen"] == 1 and rf["n_devices_distinct"] == 1
    cb = d["cpu_baseline"]
    assert list(cb)[:5] == ["value", "unit", "cores", "kind", "sample"] and cb["kind"] in ("reference", "port")
    for k in ("value", "cores", "single_thread", "beltCTR_GiBps", "bignVerify_sigs_per_s", "mixed_msgs_per_s"):
        assert isinstance(cb[k], (int, float)), k
    assert cb["cores"] <= cb["cpu_count"]
    # one figure per line before the JSON line
    assert sum(1 for l in r.stdout.splitlines() if l.startswith("[bench] ")) >= 4
    det = json.load(open(os.path.join(ROOT, d["detail"])))
    st = det["strong"]
    for w in ("bashF", "ctr", "verify", "mixed"):
        assert 1.0 < st[f"strong_pred_2_{w}"] <= 2.3 and st[f"strong_pred_2_{w}"] < st[f"strong_pred_4_{w}"] < st[f"strong_pred_8_{w}"] <= 9.0, w
        assert st[f"strong_ms_total_{w}"] > st[f"strong_ms_share8_{w}"] > 0
    assert det["others"]["bignVerify"]["verdicts_as_expected"] is True
    mr = det["others"]["bash512_beltMAC"]["roofline"]
    assert mr["bound"] == "valu-int+lds" and mr["peak"] >= mr["sum_of_parts_ceiling"] > 0 and mr["work_per_message"]["perms"] == 65
    assert det["others"]["bashF_detail"]["valu"]["clock_measured"] is True
    assert det["headline"]["host_api"]["value"] > 0
    if shutil.which("rocprofv3"):
        assert "measured in this run" in det["others"]["bashF_detail"]["traffic_source"]
        assert det["others"]["bashF_detail"]["traffic_live"]["launches"] >= 5


@pytest.mark.gpu
def test_bench_gpus_2_on_one_device_runs_two_ranks_and_explains_itself():
    """`python bench.py --gpus 2` as the driver would call it, on a one-GPU box: gloo collectives, both ranks on cuda:0.
    The line must carry what the first real 8-GPU run will need to be read: rank 0's solo rate, the weak-scaling
    efficiency against it (two ranks sharing ONE card: about 0.5 -- the documented value of this rehearsal), each rank's
    own rate, the number of distinct devices (1 here) and the clocks."""
    r = _run(["--gpus", "2", "--steps", "10", "--warmup", "2", "--ctr-gib", "1"], timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = _check_line_shape(_last_line(r.stdout), 2)
    rf = d["roofline"]
    assert d["n_gpus"] == 2 and rf["n_ranks_seen"] == 2 and d["scaling"] == "weak"
    assert rf["n_devices_distinct"] == 1
    # the strong split on two ranks that share ONE card: the same total in two halves side by side -- about 1x (the documented value
    # of this rehearsal; 2x needs a second card); what matters here is that every workload reports it and the rates are consistent
    st = json.load(open(os.path.join(ROOT, d["detail"])))["strong"]
    for w in ("bashF", "ctr", "verify", "mixed"):
        assert 0.5 < rf[f"strong_speedup_{w}"] < 2.2, (w, rf[f"strong_speedup_{w}"])
        assert abs(st[f"strong_value_{w}"] / st[f"strong_solo_value_{w}"] - st[f"strong_speedup_{w}"]) < 1e-6
    rk = d["ranks"]
    assert rk["solo_value"] > 1e9 and 0.35 < rf["weak_efficiency"] < 0.75, rf["weak_efficiency"]
    assert 0 < rk["per_rank_value_min"] <= rk["per_rank_value_max"]
    assert rk["clock_ghz_min"] and rk["clock_ghz_max"]
    assert rf["mixed_frac"] and rf["bignVerify_frac"]
    assert d["cpu_baseline"]["value"] is None and "N=1 only" in d["cpu_baseline"]["sample"]


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_strong_split_arithmetic_and_single_rank_keys():
    """the N = 1 line predicts the fixed-N split from one GPU -- strong_pred_G_w = t(total) / t(total / G)"""
    b = _bench_module()
    assert b.STRONG_TOTALS == {"bashF": 1 << 20, "ctr": 1 << 30, "verify": 1 << 18, "mixed": 1 << 24}
    assert b.strong_shares(1 << 18) == {2: 1 << 17, 4: 1 << 16, 8: 1 << 15}
    assert b.strong_shares(1000, (3,)) == {3: 333}                       # shard_range's first share
    pred = b.strong_pred(2.17, {2: 1.20, 4: 0.875, 8: 0.545})            # round 4's verification sweep
    assert abs(pred[8] - 3.98) < 0.01 and abs(pred[4] - 2.48) < 0.01 and abs(pred[2] - 1.81) < 0.01
    r = _run(["--gpus", "1", "--launch-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert set(d["strong_keys"]) == {f"strong_pred_{g}_{w}" for g in (2, 4, 8) for w in ("bashF", "ctr", "verify", "mixed")}


def test_host_cpus_reports_what_the_process_may_use():
    """cpu_baseline.cores must be the threads the process can really run (affinity mask, cgroup quota), not os.cpu_count()"""
    b = _bench_module()
    hc = b.host_cpus()
    assert 1 <= hc["threads"] <= hc["cpu_count"] and hc["threads"] <= hc["affinity"] == len(os.sched_getaffinity(0))
    if hc["cgroup_quota_cpus"]:
        assert hc["threads"] <= max(1, int(hc["cgroup_quota_cpus"]))
    one = sorted(os.sched_getaffinity(0))[:1]
    code = ("import importlib.util,os,sys,json; sys.argv=['bench.py'];"
            f"spec=importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r});"
            "b=importlib.util.module_from_spec(spec); spec.loader.exec_module(b); print(json.dumps(b.host_cpus()))")
    r = subprocess.run(["taskset", "-c", str(one[0]), sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    if r.returncode == 0:                                                # (taskset present)
        assert json.loads(r.stdout.splitlines()[-1])["threads"] == 1


def test_oracle_pool_runs_every_slice_once_and_repeats_on_request():
    """oracle/orc_threads.c: the persistent pool behind the all-cores baseline"""
    import ctypes
    import numpy as np
    import orclib
    orc = orclib.load()
    n = 1 << 12
    a = np.arange(192 * n, dtype=np.uint64).astype(np.uint8)
    want = a.copy()
    orc.lib.orc_bashF_batch(ctypes.c_void_p(want.ctypes.data), ctypes.c_size_t(n), 1)
    for threads in (2, 3, 7):
        got = a.copy()
        orc.lib.orc_bashF_batch(ctypes.c_void_p(got.ctypes.data), ctypes.c_size_t(n), threads)
        assert (got == want).all()
    twice = want.copy()
    orc.lib.orc_bashF_batch(ctypes.c_void_p(twice.ctypes.data), ctypes.c_size_t(n), 1)
    try:
        orc.lib.orc_set_slice_reps(2)
        got = a.copy()
        orc.lib.orc_bashF_batch(ctypes.c_void_p(got.ctypes.data), ctypes.c_size_t(n), 5)
    finally:
        orc.lib.orc_set_slice_reps(1)
    assert (got == twice).all()
