#!/usr/bin/env python3
"""bench.py -- bee2 hot path on MI355X: bashF perms/s (headline), beltCTR GiB/s, bign-curve256v1 verifies/s,
bash512+beltMAC messages/s (BASELINE.json configs[1..4]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--all | --only a,b,...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM: the
headline step is bashF over 2^20 independent 192-byte states.  Batches are independent, so ranks shard by index with no data-path
collective ("weak" scaling: per-GPU work fixed); RCCL carries the one parameter broadcast (key / ctr0, 48 bytes) and the
max-over-ranks of each timed region.

Output (bench_legs/line.py): one figure per line for a reader, then -- LAST line of stdout, rank 0 -- ONE strict-JSON line of at
most 8000 characters: the contract's fields, `roofline` (24 flat scalars: the headline kernel's algorithmic bytes / hipEvent
launch time against the HBM peak, the other three BASELINE metrics with their fractions, the fixed-N (strong) split of SURVEY 8e
as strong_pred_8_* at N = 1 / strong_speedup_* at N > 1, ranks / devices seen; `traffic` = the PMC HBM bytes of the headline launch,
taken in this very run by re-running that leg under rocprofv3 --pmc, bench_legs/pmc_live.py) and `cpu_baseline` (the reference itself,
oracle/_ref, on this box's host cores; rank 0, N = 1).  Every leg's full record goes to gpurun_out/bench_detail.json.

Default: the four BASELINE legs.  --all adds the SURVEY 8f legs (signing, bulk modes, ragged hashes, belt-dwp, the wider curves,
key validation / key-table verification, single-call latency); --only picks legs by name.  The legs live in bench_legs/.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

import torch  # noqa: E402

import bee2_amd  # noqa: E402
from bee2_amd import shard  # noqa: E402
from bench_legs import line as bench_line  # noqa: E402
from bench_legs.common import *  # noqa: E402,F401,F403  (constants and pure helpers: tests import them from here)
from bench_legs.common import Ctx, Dist, STRONG_TOTALS, STRONG_WAYS, LDS_CTR_CEIL_GIBPS, check_distinct_devices, device_identity  # noqa: E402
from bench_legs.cpu import host_cpus  # noqa: E402

BASELINE_LEGS = ("bashF", "ctr", "verify", "mixed")
MORE_LEGS = ("verify_more", "verify_wide", "sign", "latency", "ragged", "dwp", "modes")
WORKLOADS = BASELINE_LEGS + MORE_LEGS
LEG_MODULES = {"bashF": "bashf", "ctr": "ctr", "verify": "verify", "mixed": "mixed", "verify_more": "verify_more",
               "verify_wide": "verify_wide", "sign": "sign", "latency": "latency", "ragged": "ragged", "dwp": "dwp", "modes": "modes"}
RUN_ORDER = ("bashF", "ctr", "verify", "verify_more", "verify_wide", "sign", "latency", "mixed", "ragged", "dwp", "modes")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--all", action="store_true", help="every leg, not only the four BASELINE workloads (N = 1)")
    ap.add_argument("--only", default="", help=f"comma list of {{{','.join(WORKLOADS)}}}; default {','.join(BASELINE_LEGS)}")
    ap.add_argument("--ctr-gib", type=float, default=16.0)
    ap.add_argument("--headline-only", action="store_true",
                    help="each leg launches ONLY its BASELINE-sized batch (no strong split, no sweeps, no host-pointer legs): what "
                         "tools/profile_headline.sh profiles, so that rocprofv3's per-kernel averages are over launches of ONE size")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not re-run the headline leg under rocprofv3 --pmc for `roofline.traffic` (then the committed profile of the "
                         "same launch is replayed, or null)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="ranks only form the process group, reduce one number and rank 0 prints the line's launch fields "
                         "(no GPU work; tests/test_bench_launch.py runs this on CPU with BEE2_BENCH_BACKEND=gloo)")
    args = ap.parse_args(argv)
    if not args.only:
        args.only = ",".join(WORKLOADS if args.all and args.gpus == 1 else BASELINE_LEGS)
    bad = set(x for x in args.only.split(",") if x) - set(WORKLOADS)
    if bad:                                # fail before any GPU work, not with an empty JSON line
        ap.error(f"unknown --only name(s) {sorted(bad)}; choose from {list(WORKLOADS)}")
    return args


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment (how the driver calls the bench): become the
    launcher -- re-run this very command line under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 --
    and pass the ranks' output and exit code through.  Returns only when this process IS a rank (or N == 1)."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    backend = os.environ.get("BEE2_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not args.launch_selftest:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s); refusing to report n_gpus={args.gpus} "
                     "(set BEE2_BENCH_BACKEND=gloo to run the N-rank code path on fewer devices)")
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] self-launch: {' '.join(cmd)}", file=sys.stderr)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    sys.exit(subprocess.call(cmd, env=env))


def launch_selftest(args):
    """No GPU work: the process group, one all-reduce, the device rule, the strong split's share arithmetic -- and a line built by
    the same bench_legs/line.py from stand-in numbers, so that the first real N-GPU run cannot fail on plumbing or on the line's
    size (tests/test_bench_launch.py)."""
    dist = Dist(args.gpus, use_cuda=False)
    seen = int(dist.sum(1.0))             # one all-reduce: every rank counted
    slowest = dist.max(float(dist.rank))
    # BEE2_BENCH_MOCK_DEVICES="0,0,1": the device each rank would report (CPU test of the distinct-device rule);
    # BEE2_BENCH_MOCK_BACKEND=nccl applies RCCL's rule (one rank per GPU) to that list
    mock = os.environ.get("BEE2_BENCH_MOCK_DEVICES")
    me = f"mock:{mock.split(',')[dist.rank]}" if mock else f"cpu:{dist.local}"
    ids = dist.gather(me)
    distinct = check_distinct_devices(ids, dist.world, os.environ.get("BEE2_BENCH_MOCK_BACKEND", dist.backend))
    # the strong split: every rank's shard_range share of each fixed BASELINE job, summed over the ranks, must be the job
    covered = {w: int(dist.sum(float(shard.shard_range(dist.rank, dist.world, t)[1] - shard.shard_range(dist.rank, dist.world, t)[0])))
               for w, t in STRONG_TOTALS.items()}
    if dist.rank == 0:
        N = dist.world
        # stand-in numbers of the real magnitudes, every optional field present: the longest line the real run can print
        big = 12345678901.234567
        result = {"metric": "bashF perms/s", "value": big * N, "unit": "perms/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": 0.0912345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                  "data": "synthetic", "config": {"workload": "launch selftest (no GPU work): the line's shape with stand-in numbers", "states_per_gpu": 1 << 20,
                                                  "parallelism": f"dp{N}"},
                  "roofline": {"kernel": "bashF_tile_kernel<0, 2, 124, 6, 3>", "bound": "hbm", "achieved": 4321.123456, "peak": 8000.0, "unit": "GB/s",
                               "frac": 0.54014, "traffic": 402653184.0, "avg_launch_ms": 0.0931234, "frac_2p22": 0.51234, "valu_busy": 1.7512345}}
        others = {"beltCTR": {"value": 963.123456 * N, "roofline": {"frac": 0.2612345}},
                  "bignVerify": {"value": 124412345.678 * N, "roofline": {"frac": 0.4512345}},
                  "bash512_beltMAC": {"value": 103123456.78 * N, "roofline": {"frac": 0.5612345}}}
        strong = ({f"strong_pred_{g}_{w}": 1.234567 * g for g in STRONG_WAYS for w in STRONG_TOTALS} if N == 1
                  else {f"strong_speedup_{w}": 0.87654321 * N for w in STRONG_TOTALS})
        diag = {"weak_efficiency": 0.987654321, "solo_value": big, "per_rank_value_min": big, "per_rank_value_max": big,
                "clock_ghz_min": 1.7654321, "clock_ghz_max": 1.8765432} if N > 1 else {}
        if N == 1:
            result["cpu_baseline"] = {"value": 151234567.89, "unit": "perms/s", "cores": 16, "kind": "reference", "impl": "bee2 BASH_AVX512",
                                      "sample": "the 2^20-state batch, 16 threads over disjoint slices; 12 timed passes, each thread its slice x 3", "single_thread": 9412345.678,
                                      "scaling_over_single_thread": 16.0678, "spin_scaling": 15.98765, "cpu_count": 256}
        ln, _ = bench_line.build(result, others, strong, diag, N, seen, distinct, LDS_CTR_CEIL_GIBPS, "gpurun_out/bench_detail.json")
        ln["metric"] = "launch selftest"
        ln["value"] = None
        ln["roofline"].update({f"strong_items_{w}": c for w, c in covered.items()})
        ln.update(strong_keys=list(bench_line.strong_keys(N)) if N > 1 else [f"strong_pred_{g}_{w}" for g in STRONG_WAYS for w in STRONG_TOTALS],
                  max_rank=int(slowest), backend=dist.backend, only=args.only)
        print(bench_line.dumps(ln))
    dist.close()


def main():
    args = parse()
    self_launch(args)                         # --gpus N > 1 without RANK: does not return
    if args.launch_selftest:
        return launch_selftest(args)
    dist = Dist(args.gpus)
    eng = bee2_amd.load()                     # fails loudly without libbee2hip.so
    eng.set_device(torch.cuda.current_device())
    only = [w for w in RUN_ORDER if w in set(x for x in args.only.split(",") if x)]
    c = Ctx(args, dist, eng)
    N = c.N
    dev_ids = dist.gather(device_identity(dist.device))
    n_distinct = check_distinct_devices(dev_ids, N, dist.backend)     # RCCL: N ranks on fewer than N GPUs is an error
    c.do_cpu = (not args.no_cpu) and dist.rank == 0 and N == 1
    c.hc = host_cpus()                        # cpu_count / affinity / cgroup quota: threads = what this process may really use
    c.cores = c.hc["threads"]
    c.H = eng.beltH()
    # the only cross-GPU payload: expanded key (32 B) + ctr0 (16 B), broadcast once over RCCL
    if dist.rank == 0:
        kw, c0 = eng.beltCTRStart(c.H[128:160], c.H[192:208])      # ctr0 = E_K(iv) on the GPU
    else:
        kw, c0 = bytes(32), bytes(16)
    blob = dist.bcast_bytes(kw + c0, 48)
    c.kw, c.c0 = blob[:32], blob[32:]

    import importlib
    for w in only:
        if w in ("latency",) and not (dist.rank == 0 and N == 1):
            continue
        importlib.import_module(f"bench_legs.{LEG_MODULES[w]}").run(c)
        torch.cuda.empty_cache()

    result, others = c.result, c.others
    # HBM traffic of the headline launch, measured now (rank 0, N = 1; outside every timed region): bench_legs/pmc_live.py
    if (dist.rank == 0 and N == 1 and "bashF" in only and not args.no_live_pmc and not args.headline_only
            and isinstance(result.get("roofline"), dict)):
        from bench_legs import pmc_live
        live = pmc_live.headline_traffic("bashF", "bashF_tile_kernel", 1 << 20)
        if live:
            result["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            if live.get("valu_busy") is not None:
                result["roofline"]["valu_busy"] = live["valu_busy"]
            others.setdefault("bashF_detail", {})["traffic_source"] = live["source"]
            others["bashF_detail"]["traffic_live"] = live
        else:
            others.setdefault("bashF_detail", {})["traffic_live"] = None     # (the replayed figure, if any, stays; its source is named beside it)
    if not result:                        # --only without bashF: promote the first other metric
        k0 = next(iter(others))
        o = others.pop(k0)
        result.update({"metric": o.get("metric", k0), "value": o.get("value"), "unit": o.get("unit"), "n_gpus": N, "steps": o.get("steps", c.K),
                       "warmup": c.W, "ms_per_step": o.get("ms_per_step"), "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": o.get("config", {"workload": k0})})
        for k, v in o.items():            # keep the workload's own fields (roofline, cpu_baseline, extras)
            result.setdefault(k, v)
    seen = int(dist.sum(1.0))
    ln, detail = bench_line.build(result, others, c.strong, c.diag, N, seen, n_distinct, LDS_CTR_CEIL_GIBPS, "gpurun_out/bench_detail.json")
    detail["host"] = {"cpus": c.hc, "device": torch.cuda.get_device_name(torch.cuda.current_device()), "engine": eng.version(),
                      "n_devices_visible": dist.ndev, "argv": sys.argv[1:]}
    if dist.rank == 0:
        if bench_line.write_detail(detail, ROOT) is None:
            ln.pop("detail", None)
        for f in bench_line.figures(ln, detail):
            print(f)
        print(bench_line.dumps(ln), flush=True)
    dist.close()


if __name__ == "__main__":
    main()
