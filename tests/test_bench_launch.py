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
    assert "torch.distributed.run" in r.stderr and f"--nproc-per-node={n}" in r.stderr


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


FLAT_ROOFLINE = ("frac", "frac_2p22", "valu_frac", "n_ranks_seen", "beltCTR_GiBps", "beltCTR_frac", "beltCTR_lds_frac",
                 "bignVerify_sigs_per_s", "bignVerify_frac")


@pytest.mark.gpu
def test_bench_line_keeps_all_three_metrics_in_flat_keys():
    """the driver's record keeps `roofline` and `cpu_baseline` but drops nested objects: the beltCTR and bignVerify
    figures must be flat scalars there (short run: 1 GiB stream, 3 steps, 2 s CPU legs are part of the same code)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "BEE2_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--only", "bashF,ctr,verify", "--ctr-gib", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["metric"] == "bashF perms/s" and d["n_gpus"] == 1
    for k in FLAT_ROOFLINE:
        assert isinstance(d["roofline"][k], (int, float)), k
    assert d["roofline"]["n_ranks_seen"] == 1 and d["roofline"]["bignVerify_verdicts_ok"] is True
    assert 0 < d["roofline"]["frac"] < 1 and 0 < d["roofline"]["beltCTR_frac"] < 1 and 0 < d["roofline"]["bignVerify_frac"] < 1
    for k in ("value", "cores", "beltCTR_GiBps", "bignVerify_sigs_per_s"):
        assert isinstance(d["cpu_baseline"][k], (int, float)), k


@pytest.mark.gpu
def test_bench_gpus_2_on_one_device_runs_two_ranks():
    """`python bench.py --gpus 2` as the driver would call it, on a one-GPU box: gloo collectives, both ranks on cuda:0"""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--only", "bashF,ctr", "--ctr-gib", "1"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["roofline"]["n_ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["cpu_baseline"]["value"] is None and "N=1 only" in d["cpu_baseline"]["sample"]
