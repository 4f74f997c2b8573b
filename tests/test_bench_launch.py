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


# what the driver's record must keep: among the FIRST 16 keys of `roofline` (it cuts the object after ~24 entries)
FIRST_16 = ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_2p22", "beltCTR_GiBps", "beltCTR_frac", "beltCTR_lds_frac",
            "bignVerify_sigs_per_s", "bignVerify_frac", "mixed_msgs_per_s", "mixed_frac", "n_ranks_seen", "n_devices_distinct")
NEXT_8 = ("weak_efficiency", "solo_value", "per_rank_value_min", "per_rank_value_max", "clock_ghz_min", "clock_ghz_max",
          "avg_launch_ms", "kernel")


@pytest.mark.gpu
def test_bench_line_keeps_every_fraction_in_the_first_flat_keys():
    """the driver's record keeps `roofline` and `cpu_baseline` but drops nested objects and everything after ~24 entries:
    the three BASELINE rates, configs[4]'s rate and all their fractions must be flat scalars among the first 16 keys
    (short run: 1 GiB stream, 3 steps, 2 s CPU legs are part of the same code)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "BEE2_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--only", "bashF,ctr,verify,mixed", "--ctr-gib", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["metric"] == "bashF perms/s" and d["n_gpus"] == 1
    keys = list(d["roofline"])
    assert tuple(keys[:16]) == FIRST_16 and tuple(keys[16:24]) == NEXT_8, keys[:24]
    for k in FIRST_16:
        if k not in ("bound", "unit"):
            assert isinstance(d["roofline"][k], (int, float)), k
    for k, v in d["roofline"].items():                 # flat: no nested object, no prose beyond the kernel's name
        assert not isinstance(v, (dict, list)), k
        assert not isinstance(v, str) or len(v) <= 60, k
    rf = d["roofline"]
    assert rf["n_ranks_seen"] == 1 and rf["n_devices_distinct"] == 1 and rf["bignVerify_verdicts_ok"] is True
    assert rf["weak_efficiency"] is None and rf["solo_value"] is None           # N = 1: nothing to compare with
    for k in ("frac", "beltCTR_frac", "beltCTR_lds_frac", "bignVerify_frac", "mixed_frac", "mixed_frac_sum_of_parts"):
        assert 0 < rf[k] < 1.05, (k, rf[k])
    assert rf["clock_ghz_min"] and 1.0 < rf["clock_ghz_min"] <= rf["clock_ghz_max"] < 2.6
    assert rf["per_rank_value_min"] == rf["per_rank_value_max"] > 1e9
    # configs[4] has a roofline of its own now
    mr = d["others"]["bash512_beltMAC"]["roofline"]
    assert mr["bound"] == "valu-int+lds" and mr["peak"] >= mr["sum_of_parts_ceiling"] > 0 and mr["work_per_message"]["perms"] == 65
    # the VALU figure is labelled as the model it is, at the measured clock
    assert "valu_frac" not in rf and d["others"]["bashF_detail"]["valu"]["clock_measured"] is True
    for k in ("value", "cores", "beltCTR_GiBps", "bignVerify_sigs_per_s", "mixed_msgs_per_s"):
        assert isinstance(d["cpu_baseline"][k], (int, float)), k


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
    d = json.loads(lines[0])
    rf = d["roofline"]
    assert d["n_gpus"] == 2 and rf["n_ranks_seen"] == 2 and d["scaling"] == "weak"
    assert rf["n_devices_distinct"] == 1
    assert tuple(list(rf)[:16]) == FIRST_16 and tuple(list(rf)[16:24]) == NEXT_8
    assert rf["solo_value"] > 1e9 and 0.35 < rf["weak_efficiency"] < 0.75, rf["weak_efficiency"]
    assert 0 < rf["per_rank_value_min"] <= rf["per_rank_value_max"]
    assert rf["clock_ghz_min"] and rf["clock_ghz_max"]
    # the N > 1 default is the four BASELINE workloads
    assert set(d["others"]) >= {"beltCTR", "bignVerify", "bash512_beltMAC"} and "single_call_latency_us" not in d["others"]
    assert rf["mixed_frac"] and rf["bignVerify_frac"]
    assert d["cpu_baseline"]["value"] is None and "N=1 only" in d["cpu_baseline"]["sample"]
