#!/usr/bin/env python3
"""bench.py -- bee2 hot path on MI355X: bashF perms/s (headline), beltCTR GiB/s,
bign-curve256v1 verifies/s, bash512+beltMAC messages/s.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one batch of synthetic
input already resident in HBM: the headline step is bashF over 2^20 independent 192-byte
states (BASELINE.json configs[1]).  Batches are independent, so ranks shard by index with
no data-path collective ("weak" scaling: per-GPU work fixed); RCCL is used for the one
parameter broadcast (key / ctr0) and for the max-over-ranks of the timed region.

`value` is the WEAK reading (every rank the full BASELINE batch).  SURVEY.md 8e partitions a FIXED N ("GPU g of G takes
items [g N/G, (g+1) N/G)"), so the line carries the STRONG reading beside it, flat in `roofline`:
  N = 1   strong_pred_{2,4,8}_{bashF,ctr,verify,mixed} = t(BASELINE total on this GPU) / t(total / G on this GPU), every share
          timed like the headline (hipEvents around K launches): what a G-way split of the fixed job can reach at best
  N > 1   strong_speedup_{...} = rate of the N ranks over their shard_range(rank, N, total) shares / rank 0 alone on the total

Rank 0 prints ONE JSON line.  `value` = perms/s summed over all ranks.  Extra objects:
  roofline      dominant kernel (bashF_batch_kernel): algorithmic bytes / avg launch time,
                launch time measured inside this run with hipEvents on the launch stream
  cpu_baseline  the reference itself (oracle/_ref, bee2 compiled by oracle/Makefile) or the
                oracle port, timed on this box's host cores on a bounded sample (rank 0, N=1)
  others        the other two headline metrics of BASELINE.json + the H4 mixed job, each
                timed the same way (K steps, barrier + synchronize on both sides)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bee2_amd  # noqa: E402
from bee2_amd import shard  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BASHF_BYTES = 384                # algorithmic bytes per permutation (192 read + 192 written)
CTR_BYTES_PER_BLOCK = 32         # 16 read + 16 written per 16-byte block
LDS_CTR_CEIL_GIBPS = 256 * 2.17e9 / (220 * 2 / 64) * 16 / 2 ** 30   # 220 ds_read_b32 per block (round 3: 55 G-boxes) = 6.875 LDS clocks per block per CU
MADS_PER_VERIFY = 976 * 72 + 685 * 52 + 3000   # v_mad_u64_u32 per signature: affine table, shared inversion
MAD_PEAK_T = 30.0                # measured: 256 CU x 4 SIMD x 64 lanes x 2.31 GHz / 5.05 cycles
MAD_PEAK_GHZ = 2.31              # the shader clock of that micro-benchmark run (profiles/r01_valu_rates_ubench.txt)


# VALU-issue picture: a wave64 full-rate op occupies its SIMD for 2 cycles, a half-rate op
# for 4 (157 TFLOP/s fp32 = 1024 SIMDs x 2.4 GHz x 32 lanes x 2); tools/ubench/valu_rates.hip sustains
# 80 % of that.  Per-unit instruction counts are the ISA's (hipcc -S), not estimates.
N_SIMD = 1024                     # 256 CUs x 4
NOMINAL_GHZ = 2.4
BASHF_VALU = {"full_rate": 4 * 684, "half_rate": 4 * 384}          # per permutation-wavefront (24 rounds)
CTR_VALU = {"full_rate": 232, "half_rate": 293, "ds_read_b32": 220}  # per block-wavefront (loop body of beltCTR_blocks_kernel<BeltTabTwoP, 1, 7>, llvm-objdump;
#                                                                       before the one-instruction LDS addresses: 681 / 18 / 220 -- a third more instructions, fewer VALU cycles, 15 % slower)


def valu_picture(units_per_s, mix, clock_ghz=None, lanes=64):
    """The 2-cycle / 4-cycle cost MODEL of the instruction mix against the SIMD cycles there were: `model_ratio` = modelled
    issue cycles needed per second / SIMD cycles per second at the clock the chip actually ran (measured beside the kernel;
    the nominal 2.4 GHz only when no measurement exists).  It is a DESCRIPTION, not a ceiling: the half-rate and the
    full-rate unit of a SIMD overlap once the half-rate runs are issued at raised priority, so the ratio
    passes 1 for bash-f -- by how much is exactly that overlap."""
    cyc = 2 * mix["full_rate"] + 4 * mix["half_rate"]
    ghz = clock_ghz or NOMINAL_GHZ
    return {"issue_cycles_per_wave_unit": cyc, "model_ratio": units_per_s / lanes * cyc / (N_SIMD * ghz * 1e9),
            "clock_ghz_used": ghz, "clock_measured": clock_ghz is not None, "mix": mix}


# ISA instruction counts of the fused bash512 + beltMAC kernel's two halves per 4 KiB message (65 permutations, 257 block
# encryptions; llvm-objdump of hash_mac_fused_kernel<8, true, true, BeltTabWide> and of its parts): what SURVEY 8d row 4 asks for
MIXED_WORK = {"perms": 65, "blocks": 257, "valu_full_rate": 65 * 2736 + 257 * 232, "valu_half_rate": 65 * 1536 + 257 * 293,
              "ds_read_b32": 257 * 220}


def mixed_roofline(msgs_per_s_per_gpu, perms_per_s, blocks_per_s, src):
    """configs[4]: the fused kernel against its two parts measured in THIS run on the same GPU -- the bash-f kernel (VALU
    bound) and the belt block kernel (LDS-lookup bound).  sum_of_parts = what two back-to-back passes would give, i.e. no
    overlap at all; overlap = the slower part alone, i.e. the other part entirely hidden.  `frac` is against the overlap
    ceiling (the roof), `overlap_got` = share of the possible overlap the fusion realised."""
    t_hash, t_mac = MIXED_WORK["perms"] / perms_per_s, MIXED_WORK["blocks"] / blocks_per_s
    t = 1.0 / msgs_per_s_per_gpu
    t_sum, t_max = t_hash + t_mac, max(t_hash, t_mac)
    return {"bound": "valu-int+lds", "achieved": msgs_per_s_per_gpu, "peak": 1.0 / t_max, "unit": "messages/s",
            "frac": t_max / t, "traffic": None,
            "sum_of_parts_ceiling": 1.0 / t_sum, "frac_sum_of_parts": t_sum / t,
            "overlap_got": (t_sum - t) / (t_sum - t_max) if t_sum > t_max else None,
            "part_rates": {"bashF_perms_per_s": perms_per_s, "belt_blocks_per_s": blocks_per_s, "source": src},
            "work_per_message": MIXED_WORK,
            "algorithmic_bytes_per_message": 4096 + 72,
            # why the overlap ceiling is out of reach: the belt half is not only LDS look-ups, it has VALU work of its own, and VALU
            # work of two wavefronts does not overlap.  Modelled VALU issue cycles per 64 messages (2 / 4 cycles per full- / half-rate
            # instruction) against the SIMD cycles there are at the nominal clock: near 1 = the fused kernel is VALU-bound
            "valu_model_ratio_nominal_clock": msgs_per_s_per_gpu / 64 * (2 * MIXED_WORK["valu_full_rate"] + 4 * MIXED_WORK["valu_half_rate"])
                                              / (N_SIMD * NOMINAL_GHZ * 1e9)}


def device_identity(index):
    """something that names the physical GPU behind a HIP device index (two ranks on one card must not count as two): the index
    itself AND whatever the runtime knows about the card.  All parts together: a box whose cards all report the same (e.g. zero)
    uuid still counts N devices when the ranks sit on N indices, and ranks that each see one card as index 0 (a launcher that
    sets HIP_VISIBLE_DEVICES per rank) are told apart by uuid / PCI bus id."""
    parts = [f"index:{index}"]
    try:
        p = torch.cuda.get_device_properties(index)
        for attr in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
            v = getattr(p, attr, None)
            if v not in (None, ""):
                parts.append(f"{attr}:{v}")
    except Exception:
        pass
    return "|".join(parts)


def check_distinct_devices(ids, world, backend):
    """n_devices_distinct; with RCCL (one rank per GPU is the contract) anything but `world` distinct devices is an error"""
    distinct = len(set(ids))
    if backend == "nccl" and distinct != world:
        raise SystemExit(f"[bench] {world} ranks on {distinct} distinct device(s) {sorted(set(ids))}: refusing to report "
                         f"n_gpus={world} (BEE2_BENCH_BACKEND=gloo runs the N-rank code path on fewer devices)")
    return distinct


STRONG_TOTALS = {"bashF": 1 << 20, "ctr": 1 << 30, "verify": 1 << 18, "mixed": 1 << 24}   # BASELINE configs[1..4]: states, 16-byte blocks, signatures, messages
STRONG_WAYS = (2, 4, 8)


def strong_shares(total, ways=STRONG_WAYS):
    """{G: items of rank 0's share} of a fixed job of `total` items split G ways by shard.shard_range"""
    return {g: shard.shard_range(0, g, total)[1] for g in ways}


def strong_pred(t_total_ms, t_share_ms):
    """one-GPU PREDICTION of the G-way strong speedup: {G: t(total) / t(total / G)}, both times measured on this GPU"""
    return {g: (t_total_ms / t if t else None) for g, t in t_share_ms.items()}


def event_ms(fn, steps, warmup=2, graph=False):
    """average milliseconds per call of fn(): hipEvents on the launch stream around `steps` calls, after `warmup` untimed ones.
    graph=True: the `steps` calls are captured into ONE hipGraph and the replay is timed -- for launches of a few microseconds,
    where an eager Python loop would measure the host's launch rate (2^17 bashF states: 21 us eager, 17 us on the device)."""
    # (the chip drops its clock within milliseconds of going idle and needs ~0.2 s of load to come back -- timed() pre-warms the same way;
    #  without this a 2.5 ms measurement behind an idle phase runs at the idle clock: 127 us per 2^20-state launch instead of 94)
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if graph:
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(steps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


WORKLOADS = ("bashF", "ctr", "verify", "sign", "mixed", "modes", "ragged", "dwp", "latency")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--only", default="", help="comma list of {bashF,ctr,verify,sign,mixed,modes,ragged,dwp,latency}; default all")
    ap.add_argument("--ctr-gib", type=float, default=16.0)
    ap.add_argument("--headline-only", action="store_true",
                    help="bashF: only the 2^20-state launches (no 2^22 leg, no host-pointer leg) -- what tools/profile_round.sh "
                         "profiles, so that rocprofv3's per-kernel averages are over launches of ONE size")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="ranks only form the process group, reduce one number and rank 0 prints the line's launch fields "
                         "(no GPU work; tests/test_bench_launch.py runs this on CPU with BEE2_BENCH_BACKEND=gloo)")
    args = ap.parse_args()
    if not args.only and args.gpus > 1:
        # eight ranks share one host: only the four BASELINE workloads by default (latency / ragged / modes legs are N = 1 matter)
        args.only = "bashF,ctr,verify,mixed"
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


class Dist:
    """one process per GPU; RCCL ('nccl') by default.  BEE2_BENCH_BACKEND=gloo runs the same
    code with CPU-side collectives (lets the N>1 path be exercised on a box with one GPU)."""

    def __init__(self, want, use_cuda=True):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        # under torchrun the process group is set up even for one rank, so that a single-GPU box exercises the very
        # RCCL calls (init with device_id, broadcast, all-reduce, barrier) the 2/4/8-GPU runs make
        self.on = self.world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
        self.backend = os.environ.get("BEE2_BENCH_BACKEND", "nccl")
        self.ndev = torch.cuda.device_count()
        self.device = self.local % max(1, self.ndev)
        if use_cuda:
            torch.cuda.set_device(self.device)
        if self.on:
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            else:
                dist.init_process_group(self.backend)
        self.cdev = "cuda" if self.backend == "nccl" else "cpu"
        if want != self.world:
            # never report an n_gpus that is not the number of ranks that ran
            raise SystemExit(f"[bench] --gpus {want} but WORLD_SIZE={self.world}: launch with --nproc-per-node {want} "
                             f"(or run `python bench.py --gpus {want}` and let it launch the ranks itself)")

    def sum(self, x):
        if not self.on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def barrier(self):
        if self.on:
            self.dist.barrier()

    def max(self, x):
        if not self.on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """every rank's object, in rank order, on every rank"""
        if not self.on:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def bcast_bytes(self, b, n):
        """rank 0's bytes to everyone (the only payload that crosses GPUs: <= 48 bytes)"""
        t = torch.zeros(n, dtype=torch.uint8, device=self.cdev)
        if self.rank == 0:
            t.copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
        if self.on:
            self.dist.broadcast(t, src=0)
        return t.cpu().numpy().tobytes()

    def close(self):
        if self.on:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(dist, steps, warmup, fn):
    """clock pre-warm, W untimed steps, then exactly K timed steps between
    barrier+synchronize pairs; returns the max over ranks of the elapsed seconds"""
    # the chip needs a few hundred ms of load to reach its sustained clock (DVFS); without this
    # the first launches of a short run are ~8 % slower than steady state (tools/gap_test.py)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    # hipEvents on the stream the kernels are launched on (torch's current stream: Engine._stream), recorded
    # INSIDE the timed region around the same K launches: roofline.avg_launch_ms = their distance / K
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    timed.own_wall = time.perf_counter() - t0                    # this rank alone, before it waits for the others
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    timed.event_ms = e0.elapsed_time(e1) / steps
    return dist.max(wall)


timed.event_ms = 0.0
timed.own_wall = 0.0


class _Alone:
    """stands in for Dist inside a solo leg: no collective"""
    def barrier(self):
        pass

    def max(self, x):
        return x


def solo_timed(dist, steps, warmup, fn):
    """rank 0 ALONE times the step while the other ranks wait at a barrier (N > 1 only): what one GPU of this node does
    with the host and the fabric to itself.  Returns rank 0's seconds for `steps` steps on every rank."""
    dist.barrier()
    el = 0.0
    if dist.rank == 0:
        el = timed(_Alone(), steps, warmup, fn)
    dist.barrier()
    return dist.max(el)


def clock_probe_lib():
    path = os.path.join(ROOT, "bee2_amd", "lib", "libb2hprobe.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


def shader_clock_under(fn, ms_launch):
    """GHz the chip sustained under fn(): one wavefront on a side stream spins beside ~100 more launches (outside any timed
    region) and reads s_memtime against the 100 MHz s_memrealtime (tools/probe/clock_probe.hip).  None without the helper."""
    lib = clock_probe_lib()
    if lib is None:
        return None, None
    side = torch.cuda.Stream()
    probe = torch.zeros(2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(20):                                   # the queue is already full when the probe arrives
        fn()
    if lib.b2h_clock_probe(ctypes.c_void_p(probe.data_ptr()), ctypes.c_uint(max(50, int(ms_launch * 1e3 * 40 * 0.8))),
                           ctypes.c_void_p(side.cuda_stream)) != 0:
        return None, None
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    c = probe.cpu().numpy()
    under = float(c[0]) / (float(c[1]) * 10.0)
    lib.b2h_clock_probe(ctypes.c_void_p(probe.data_ptr()), ctypes.c_uint(2000), ctypes.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()
    c = probe.cpu().numpy()
    return under, float(c[0]) / (float(c[1]) * 10.0)


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc summary of this round (separate FETCH_SIZE /
    WRITE_SIZE passes, FETCH_SIZE doubled for wide coalesced reads as MI355X_MICROARCH.md prescribes): a replay of
    that profiling run, labelled as such -- counters cannot be collected inside an un-profiled bench run."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_summary.json")))
    if not found:
        return None, None
    path = found[-1]                                       # the newest round's counters
    try:
        d = json.load(open(path))
        for k, v in d.get("kernels", {}).items():
            if kernel_substr in k and "hbm_bytes_per_launch" in v:
                return v["hbm_bytes_per_launch"], f"profiles/{os.path.basename(path)} @ {d.get('commit', '?')} ({v.get('note', 'rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE')})"
    except Exception:
        pass
    return None, None


def pmc_valu(kernel_substr):
    """valu_busy of a kernel from the newest committed rocprofv3 --pmc summary: a REPLAY of that profiling run, like `traffic` --
    north_star's "VALU integer-op utilisation" beside each fraction.  valu_busy = SQ_ACTIVE_INST_VALU x 4 / SIMD cycles = the average
    number of VALU instructions EXECUTING per SIMD: 1.0 = one pipe never idle (the multiply-add / carry kernels, whose classes do not
    overlap), up to 2.0 where half-rate and full-rate instructions of different wavefronts run side by side (bash-f: 1.75)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_summary.json")))
    for path in reversed(found):
        try:
            d = json.load(open(path))
            for k, v in d.get("kernels", {}).items():
                if kernel_substr in k and "valu_busy" in v:
                    return v["valu_busy"]
        except Exception:
            pass
    return None


# ------------------------------------------------------------------------- CPU baseline

def fill_seeded(t, seed):
    """synthetic input generated in HBM by torch's counter-based generator (Philox) with the seed
    SURVEY.md 8d assigns to the workload (+ rank, so ranks hold different data)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    t.view(torch.int64).random_(generator=g)


def host_api_rate(call, units, reps=3):
    """PCIe-inclusive rate of a host-pointer C-ABI entry (H2D + kernels + D2H inside the call, host
    buffers in ordinary pageable memory, as a C caller of the drop-in would have).  Reported beside
    `value`, never as `value`."""
    call()                                                  # first call pays allocation / table set-up
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    return units / dt, dt * 1e3

def host_cpus():
    """How many host threads this process may really use: os.cpu_count() is the machine, the affinity mask and the cgroup CPU
    quota are this process's share of it.  threads = min of the three; all of them are reported."""
    count = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = count
    quota = None
    try:                                                   # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        try:                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    threads = max(1, min(count, aff, int(quota) if quota and quota >= 1 else count))
    phys = None
    try:                                                   # physical cores behind the logical ones (SMT siblings share an ALU)
        pairs, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip() and cur:
                pairs.add((cur.get("physical id"), cur.get("core id")))
                cur = {}
        phys = len(pairs) or None
    except Exception:
        pass
    return {"cpu_count": count, "affinity": aff, "cgroup_quota_cpus": quota, "threads": threads, "physical_cores": phys}


def spin_scaling(orc, threads):
    """measured: total rate of `threads` threads each running a dependent 64-bit multiply-add chain / the rate of one thread
    (oracle/orc_threads.c orc_spin_rate): how many cores' worth of cycles the box really gives this process"""
    orc.lib.orc_spin_rate.restype = ctypes.c_double
    orc.lib.orc_spin_rate(int(threads), ctypes.c_double(0.5))    # untimed: the pool's threads are created here, and freshly created
    #                                                              threads take a few hundred ms to spread over the CPUs
    r1 = orc.lib.orc_spin_rate(1, ctypes.c_double(0.4))
    # best of three: the first pass after a single-thread phase can run with the woken threads still queued on one CPU
    rt = max(orc.lib.orc_spin_rate(int(threads), ctypes.c_double(0.5)) for _ in range(3))
    return rt / r1 if r1 else None


def cpu_baseline(which, hc):
    """Time the reference (or the oracle port) on one host thread and on hc["threads"] threads of the oracle's persistent pool,
    every thread >= ~100 ms of work per timed pass (its slice repeated: orc_set_slice_reps).  Returns dict."""
    import orclib
    import refgen
    orc = orclib.load()
    cores = hc["threads"]
    cpuflags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    have_avx512 = " avx512f" in cpuflags
    ref = None
    kind = "port"
    variant = "oracle scalar C"
    if refgen.have_ref():
        path = refgen.REF_AVX512_SO if (which == "bashF" and have_avx512 and os.path.exists(refgen.REF_AVX512_SO)) else refgen.REF_SO
        ref = ctypes.CDLL(path)
        kind = "reference"
        variant = "bee2 BASH_AVX512" if path == refgen.REF_AVX512_SO else "bee2 BASH_64 / scalar C"
    fnptr = lambda name: ctypes.cast(getattr(ref, name), ctypes.c_void_p)  # noqa: E731
    H = orc.beltH()
    if "spin" not in hc:
        hc["spin"] = spin_scaling(orc, cores)
    out = {"cores": cores, "kind": kind, "impl": variant, "cpu_count": hc["cpu_count"], "affinity": hc["affinity"],
           "cgroup_quota_cpus": hc["cgroup_quota_cpus"], "physical_cores": hc["physical_cores"], "spin_scaling": hc["spin"]}

    def clock(run, units, min_s=2.0, max_reps=64):
        reps, t0 = 0, time.perf_counter()
        while True:
            run()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= min_s or reps >= max_reps:
                return units * reps / dt, reps

    def both(run, n, min_s=4.0):
        """run(threads) over n units -> (all-threads rate, single-thread rate, note).  A short untimed single-thread probe sizes
        the slice repetitions so that a pass is >= ~100 ms per thread in both legs."""
        orc.lib.orc_set_slice_reps(1)
        run(1)                                             # warm caches / lazy init (the reference's curve object)
        t0 = time.perf_counter()
        run(1)
        t_unit = (time.perf_counter() - t0) / n            # seconds per unit on one thread
        reps1 = max(1, int(0.1 / max(t_unit * n, 1e-9)) + 1)
        repsN = max(1, int(0.1 / max(t_unit * n / cores, 1e-9)) + 1)
        try:
            orc.lib.orc_set_slice_reps(reps1)
            v1, _ = clock(lambda: run(1), n * reps1, min_s=1.0, max_reps=8)
            orc.lib.orc_set_slice_reps(repsN)
            run(cores)                                     # the pool's threads exist from here on
            vall, passes = clock(lambda: run(cores), n * repsN, min_s=min_s)
        finally:
            orc.lib.orc_set_slice_reps(1)
        scal = vall / v1
        note = None
        if scal < 0.7 * cores:
            note = (f"{cores} threads give {scal:.1f}x one thread; a dependent-multiply spin loop on the same pool gives "
                    f"{hc['spin']:.1f}x: that is what the box lets this process have (SMT siblings / shared vCPUs / clocks), "
                    "not a property of the code")
        return vall, v1, scal, f"{passes} timed passes, each thread its slice x {repsN} (>= 100 ms per thread per pass), persistent pool", note

    if which == "bashF":
        n = 1 << 20
        buf = np.empty(192 * n, dtype=np.uint8)
        orc.fill_np(buf, 0xBA5F)
        p = ctypes.c_void_p(buf.ctypes.data)
        if ref is not None:
            run = lambda th: orc.lib.orc_drive_ref_bashF(fnptr("bashF"), p, ctypes.c_size_t(n), th)  # noqa: E731
        else:
            run = lambda th: orc.lib.orc_bashF_batch(p, ctypes.c_size_t(n), th)  # noqa: E731
        vall, v1, scal, how, note = both(run, n)
        out.update(value=vall, unit="perms/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"the 2^20-state batch, {cores} threads over disjoint slices; {how}", scaling_note=note)
        if ref is not None and variant != "bee2 BASH_64 / scalar C":
            ref64 = ctypes.CDLL(refgen.REF_SO)
            f64 = ctypes.cast(ref64.bashF, ctypes.c_void_p)
            v64, v64_1, s64, _, _ = both(lambda th: orc.lib.orc_drive_ref_bashF(f64, p, ctypes.c_size_t(n), th), n, min_s=2.0)
            out["bash64_all_cores"] = v64
            out["bash64_single_thread"] = v64_1
            out["bash64_scaling_over_single_thread"] = s64
    elif which == "ctr":
        nbytes = 256 << 20
        buf = np.zeros(nbytes, dtype=np.uint8)
        kw, c0 = orc.ctr_start(H[128:160], H[192:208])
        p = ctypes.c_void_p(buf.ctypes.data)
        nb = nbytes // 16
        if ref is not None:
            run = lambda th: orc.lib.orc_drive_ref_ctr(fnptr("beltCTRStepE"), p, ctypes.c_size_t(nb), kw, c0, ctypes.c_uint64(0), th)  # noqa: E731
        else:
            run = lambda th: orc.lib.orc_beltCTR_blocks(p, ctypes.c_size_t(nb), kw, c0, ctypes.c_uint64(0), th)  # noqa: E731
        # (the single-thread leg runs over a 16 MiB prefix: 256 MiB would be 1.3 s per pass)
        nb1 = nb // 16
        run1 = lambda th: (orc.lib.orc_drive_ref_ctr(fnptr("beltCTRStepE"), p, ctypes.c_size_t(nb1 if th == 1 else nb), kw, c0, ctypes.c_uint64(0), th)  # noqa: E731
                           if ref is not None else orc.lib.orc_beltCTR_blocks(p, ctypes.c_size_t(nb1 if th == 1 else nb), kw, c0, ctypes.c_uint64(0), th))
        orc.lib.orc_set_slice_reps(1)
        run1(1)
        v1, _ = clock(lambda: run1(1), nb1 * 16 / 2 ** 30, min_s=1.5, max_reps=16)
        t_unit = 1.0 / (v1 * 2 ** 30 / 16)                 # seconds per block, one thread
        repsN = max(1, int(0.1 / (t_unit * nb / cores)) + 1)
        try:
            orc.lib.orc_set_slice_reps(repsN)
            run(cores)
            vall, passes = clock(lambda: run(cores), nbytes * repsN / 2 ** 30, min_s=4.0)
        finally:
            orc.lib.orc_set_slice_reps(1)
        scal = vall / v1
        out.update(value=vall, unit="GiB/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"{passes} timed passes over a 256 MiB prefix of the stream, {cores} threads, each its slice x {repsN}; one thread: a 16 MiB prefix",
                   scaling_note=None if scal >= 0.7 * cores else f"{scal:.1f}x over one thread against {hc['spin']:.1f}x for a spin loop on the same pool")
    elif which == "verify":
        G = orclib.Golden()
        hs, ss, ps = G.bign_base_arrays()
        nbase = len(hs) // 32
        tile = max(1, min(64, (cores * 64 + nbase - 1) // nbase))   # >= 64 signatures per thread: 2048 x tile
        hs, ss, ps = hs * tile, ss * tile, ps * tile
        n = nbase * tile
        codes = (ctypes.c_uint32 * n)()
        if ref is not None:
            ref.bign128Verify.restype = ctypes.c_uint32
            run = lambda th: orc.lib.orc_drive_ref_verify(fnptr("bign128Verify"), hs, ss, ps, ctypes.c_size_t(n if th > 1 else 256), codes, th)  # noqa: E731
        else:
            run = lambda th: orc.lib.orc_bign128Verify_batch(hs, ss, ps, ctypes.c_size_t(n if th > 1 else 256), codes, th)  # noqa: E731
        orc.lib.orc_set_slice_reps(1)
        run(1)
        v1, _ = clock(lambda: run(1), 256, min_s=1.0, max_reps=16)
        repsN = max(1, int(0.1 / (n / cores / v1)) + 1)
        try:
            orc.lib.orc_set_slice_reps(repsN)
            run(cores)
            vall, passes = clock(lambda: run(cores), n * repsN, min_s=4.0)
        finally:
            orc.lib.orc_set_slice_reps(1)
        assert all(c == 0 for c in codes)
        scal = vall / v1
        out.update(value=vall, unit="verifies/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"{passes} timed passes over the 2048 genuine signatures of tests/golden/bign_base.bin tiled x {tile}, {cores} threads, "
                          f"each its slice x {repsN} (>= 100 ms per thread per pass); one thread: the first 256",
                   scaling_note=None if scal >= 0.7 * cores else f"{scal:.1f}x over one thread against {hc['spin']:.1f}x for a spin loop on the same pool")
    elif which == "mixed":
        ml = 4096
        n = max(1 << 12, min(1 << 16, cores * 64))         # >= 64 messages per thread
        msgs = orc.fill(n * ml, 0x4D1C)
        dig = ctypes.create_string_buffer(64 * n)
        tag = ctypes.create_string_buffer(8 * n)
        key = H[128:160]
        if ref is not None:
            run = lambda th: orc.lib.orc_drive_ref_mixed(fnptr("bashHash"), fnptr("beltMAC"), msgs, ctypes.c_size_t(ml), ctypes.c_size_t(n if th > 1 else 256), key, ctypes.c_size_t(32), dig, tag, th)  # noqa: E731
        else:
            run = lambda th: orc.lib.orc_bash512_beltMAC_batch(msgs, ctypes.c_size_t(ml), ctypes.c_size_t(n if th > 1 else 256), key, ctypes.c_size_t(32), dig, tag, th)  # noqa: E731
        orc.lib.orc_set_slice_reps(1)
        run(1)
        v1, _ = clock(lambda: run(1), 256, min_s=1.0, max_reps=64)
        repsN = max(1, int(0.1 / (n / cores / v1)) + 1)
        try:
            orc.lib.orc_set_slice_reps(repsN)
            run(cores)
            vall, passes = clock(lambda: run(cores), n * repsN, min_s=4.0)
        finally:
            orc.lib.orc_set_slice_reps(1)
        scal = vall / v1
        out.update(value=vall, unit="messages/s", single_thread=v1, scaling_over_single_thread=scal,
                   sample=f"{passes} timed passes over {n} x 4 KiB messages, {cores} threads, each its slice x {repsN}; one thread: the first 256",
                   scaling_note=None if scal >= 0.7 * cores else f"{scal:.1f}x over one thread against {hc['spin']:.1f}x for a spin loop on the same pool")
    return out


# --------------------------------------------------------------------------------- main
def main():
    args = parse()
    self_launch(args)                         # --gpus N > 1 without RANK: does not return
    if args.launch_selftest:
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
            print(json.dumps({"metric": "launch selftest", "n_gpus": dist.world,
                              "roofline": {"n_ranks_seen": seen, "n_devices_distinct": distinct,
                                           **{f"strong_items_{w}": c for w, c in covered.items()}},
                              "strong_keys": [f"strong_speedup_{w}" for w in STRONG_TOTALS] if dist.world > 1
                                             else [f"strong_pred_{g}_{w}" for g in STRONG_WAYS for w in STRONG_TOTALS],
                              "max_rank": int(slowest), "backend": dist.backend, "only": args.only}))
        dist.close()
        return
    dist = Dist(args.gpus)
    eng = bee2_amd.load()                     # fails loudly without libbee2hip.so
    eng.set_device(torch.cuda.current_device())
    only = set(x for x in args.only.split(",") if x) or set(WORKLOADS)
    K, W, N = args.steps, args.warmup, dist.world
    dev_ids = dist.gather(device_identity(dist.device))
    n_distinct = check_distinct_devices(dev_ids, N, dist.backend)     # RCCL: N ranks on fewer than N GPUs is an error
    diag = {}                                                          # N > 1 self-explanation, flat scalars (rank 0 prints)
    do_cpu = (not args.no_cpu) and dist.rank == 0 and N == 1
    hc = host_cpus()                           # cpu_count / affinity / cgroup quota: threads = what this process may really use
    cores = hc["threads"]
    H = eng.beltH()

    # the only cross-GPU payload: expanded key (32 B) + ctr0 (16 B), broadcast once over RCCL
    if dist.rank == 0:
        kw, c0 = eng.beltCTRStart(H[128:160], H[192:208])      # ctr0 = E_K(iv) on the GPU
    else:
        kw, c0 = bytes(32), bytes(16)
    blob = dist.bcast_bytes(kw + c0, 48)
    kw, c0 = blob[:32], blob[32:]

    result = {}
    others = {}
    rates = {}
    strong = {}                                # the fixed-N (strong) reading of SURVEY 8e, flat scalars for `roofline`

    def strong_leg(name, total, unit_fn, steps, to_value=1.0, t_total_ms=None, graph=False):
        """The FIXED job of `total` items (BASELINE's size) split by shard.shard_range.  unit_fn(lo, hi) returns the step over
        items [lo, hi) of the resident job.  N = 1: time rank 0's share of a 2- / 4- / 8-way split on this GPU, like the headline
        (hipEvents around `steps` launches) -> strong_pred_G_<name> = t(total) / t(total / G).  N > 1: every rank its own
        share between barriers -> strong_value_<name> (whole job, in the metric's unit) and strong_speedup_<name> = rank 0
        alone on the total / the N ranks on their shares."""
        if N == 1:
            t_tot = t_total_ms if t_total_ms is not None and not graph else event_ms(unit_fn(0, total), steps, graph=graph)
            t_sh = {g: event_ms(unit_fn(0, m), steps, graph=graph) for g, m in strong_shares(total).items()}
            for g, v in strong_pred(t_tot, t_sh).items():
                strong[f"strong_pred_{g}_{name}"] = v
            strong[f"strong_ms_total_{name}"] = t_tot
            for g, t in t_sh.items():
                strong[f"strong_ms_share{g}_{name}"] = t
        else:
            lo, hi = shard.shard_range(dist.rank, N, total)
            solo = solo_timed(dist, steps, 2, unit_fn(0, total))          # rank 0 alone on the WHOLE job
            el_s = timed(dist, steps, 2, unit_fn(lo, hi))                 # every rank its share, max over ranks
            strong[f"strong_value_{name}"] = total * steps / el_s * to_value
            strong[f"strong_solo_value_{name}"] = total * steps / solo * to_value
            strong[f"strong_speedup_{name}"] = solo / el_s

    # ---------------------------------------------------------------- bashF (headline)
    if "bashF" in only:
        n = 1 << 20
        st = torch.empty(192 * n, dtype=torch.uint8, device="cuda")
        fill_seeded(st, 0xBA5F + dist.rank)                      # synthetic states, generated in HBM
        step = lambda: eng.bashF_batch_dev(st)  # noqa: E731
        solo_el = solo_timed(dist, K, W, step) if N > 1 else None      # rank 0 alone, before the group run
        el = timed(dist, K, W, step)
        value = N * n * K / el
        ms_launch = timed.event_ms                                # hipEvents around the K timed launches themselves
        own = dist.gather((n * K / timed.own_wall, n / (ms_launch * 1e-3)))   # each rank's own wall-clock and event rates
        ach = BASHF_BYTES * n / (ms_launch * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic("bashF_tile_kernel")
        # the shader clock the chip sustained (power: ~1.8-1.9 GHz under this kernel is normal; one box of the pool ran everything
        # at half speed): measured on every rank, outside the timed region
        try:
            ghz_under, ghz_idle = shader_clock_under(step, ms_launch)
            clock_note = None if ghz_under else "libb2hprobe.so not built"
        except Exception as e:                                    # the probe is a diagnostic, never a reason to fail the bench
            ghz_under, ghz_idle, clock_note = None, None, repr(e)
        clocks = [c for c in dist.gather(ghz_under) if c]
        result = {
            "metric": "bashF perms/s", "value": value, "unit": "perms/s", "n_gpus": N, "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "bashF batch: 2^20 independent 192-byte sponge states per GPU (BASELINE configs[1])",
                       "states_per_gpu": n, "parallelism": f"dp{N} (index-sharded, no data-path collective); `value` is the WEAK reading: every rank runs the full "
                                      "BASELINE batch; the fixed-N (strong) split of SURVEY 8e is roofline.strong_pred_* (N = 1: one-GPU prediction) "
                                      "/ roofline.strong_speedup_* (N > 1: measured)"},
            "roofline": {"kernel": "bashF_tile_kernel<0, 2, 124, 6, 3>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": ms_launch, "algorithmic_bytes_per_launch": BASHF_BYTES * n,
                         "shader_clock_ghz_under_kernel": ghz_under, "shader_clock_ghz_idle": ghz_idle},
        }
        detail = {"traffic_source": traffic_src, "shader_clock_note": clock_note,
                  "note": "power / VALU-issue bound in practice: the chip sustains 1.7-1.9 GHz under this kernel (DESIGN.md 2, 4.1)",
                  "valu": valu_picture(n / (ms_launch * 1e-3), BASHF_VALU, ghz_under)}
        others["bashF_detail"] = detail
        result["roofline"]["valu_model_ratio"] = detail["valu"]["model_ratio"]
        diag.update(per_rank_value_min=min(o[0] for o in own), per_rank_value_max=max(o[0] for o in own),
                    per_rank_kernel_rate_min=min(o[1] for o in own), per_rank_kernel_rate_max=max(o[1] for o in own),
                    clock_ghz_min=min(clocks) if clocks else None, clock_ghz_max=max(clocks) if clocks else None)
        if solo_el is not None:
            diag["solo_value"] = n * K / solo_el
            diag["weak_efficiency"] = value / (N * diag["solo_value"])
        rates["bashF_perms_per_s"] = n / (ms_launch * 1e-3)      # per GPU, kernel time: what the mixed roofline's parts use
        if not args.headline_only:
            bash_unit = lambda lo, hi: (lambda: eng.bashF_batch_dev(st[192 * lo: 192 * hi]))  # noqa: E731
            strong_leg("bashF", n, bash_unit, K, t_total_ms=ms_launch if N == 1 else None)
            if N == 1:
                # A 2^17-state share is a 17 us kernel: an eager Python loop measures the host's launch rate there (~21 us per step),
                # which is also what the N > 1 strong legs will see.  The DEVICE side of the same split: the shares as replays of a
                # captured graph of K launches (no host in the loop) against the headline's event-timed launch (device-bound at 94 us).
                for g, m in strong_shares(n).items():
                    t_dev = event_ms(bash_unit(0, m), K, graph=True)
                    strong[f"strong_ms_share{g}_bashF_device"] = t_dev
                    strong[f"strong_pred_{g}_bashF_device"] = ms_launch / t_dev
        # the same kernel on a batch that cannot sit in the 256 MiB Infinity Cache: 2^22 states = 768 MiB read + written
        # per launch; reported as flat keys next to the cache-resident headline
        if not args.headline_only:
            n22 = 1 << 22
            st22 = torch.empty(192 * n22, dtype=torch.uint8, device="cuda")
            fill_seeded(st22, 0xBA5F + 0x22 + dist.rank)
            timed(dist, max(3, min(K, 20)), 2, lambda: eng.bashF_batch_dev(st22))
            ms22 = timed.event_ms
            result["roofline"]["ms_2p22"] = ms22
            result["roofline"]["frac_2p22"] = BASHF_BYTES * n22 / (ms22 * 1e-3) / 1e9 / HBM_PEAK_GBS
            del st22
        if dist.rank == 0 and N == 1 and not args.headline_only:       # PCIe-inclusive rate: single-GPU runs only
            host = st.cpu().numpy()                               # pageable host copy of the same batch
            hp = ctypes.c_void_p(host.ctypes.data)
            v, ms = host_api_rate(lambda: eng._check(eng.lib.bee2hip_bashF_batch(hp, ctypes.c_size_t(n)), "bashF_batch"), n)
            result["host_api"] = {"entry": "bee2hip_bashF_batch", "value": v, "unit": "perms/s", "ms_per_call": ms,
                                  "note": "host pointers, PCIe both ways inside the call; 1 GPU; not `value`"}
            del host
        if do_cpu:
            result["cpu_baseline"] = cpu_baseline("bashF", hc)
        del st

    # ------------------------------------------------------------------------ beltCTR
    if "ctr" in only:
        nbytes = int(args.ctr_gib * (1 << 30)) // 16 * 16
        free, _ = torch.cuda.mem_get_info()
        if free < nbytes + (1 << 30):
            nbytes = (int(free * 0.5) // (1 << 20)) << 20
        nbytes = int(round(-dist.max(-float(nbytes))))             # one stream length for all ranks (the strong leg cuts ONE stream)
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        fill_seeded(buf, 0xBE17 + dist.rank)
        nb = nbytes // 16
        _, _, first = shard.ctr_shard(dist.rank, N, nbytes * N)     # rank r owns blocks [r nb, (r+1) nb)
        kc = max(3, min(K, 10))
        el = timed(dist, kc, 2, lambda: eng.beltCTR_blocks_dev(buf, kw, c0, first))
        ms_launch = timed.event_ms
        ach = CTR_BYTES_PER_BLOCK * nb / (ms_launch * 1e-3) / 1e9
        ctr_traffic, ctr_traffic_src = pmc_traffic("beltCTR_blocks_kernel")
        others["beltCTR"] = {
            "metric": "beltCTR GiB/s", "value": N * nbytes * kc / el / 2 ** 30, "unit": "GiB/s", "steps": kc,
            "ms_per_step": el / kc * 1e3,
            "config": {"workload": f"beltCTR bulk encrypt, {nbytes / 2**30:.1f} GiB stream per GPU, one key (BASELINE configs[2])"},
            "roofline": {"kernel": "beltCTR_blocks_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": ctr_traffic, "traffic_source": ctr_traffic_src,
                         "avg_launch_ms": ms_launch,
                         "note": "LDS-lookup / instruction-issue bound, not HBM: per block 220 ds_read_b32 (7 LDS clocks per block per CU: "
                                 "`beltCTR_lds_frac`) and ~525 VALU instructions (8 per G-box since the LDS addresses are one SDWA "
                                 "move each: profiles/r03_belt_sdwa_ab.txt; ~700 before, 10.0 CU-cycles per block then, ~8.4 now), "
                                 "DESIGN.md 2 and 4.2",
                         "valu": valu_picture(nb / (ms_launch * 1e-3), CTR_VALU)},
        }
        rates["belt_blocks_per_s"] = nb / (ms_launch * 1e-3)       # per GPU, kernel time
        if not args.headline_only:
            # the fixed job: ONE stream of nb blocks; rank r encrypts blocks [lo, hi) with first_block = lo (no state passes between ranks)
            strong_leg("ctr", nb, lambda lo, hi: (lambda: eng.beltCTR_blocks_dev(buf[16 * lo: 16 * hi], kw, c0, lo)), kc,
                       to_value=16 / 2 ** 30, t_total_ms=ms_launch if N == 1 else None)
        if dist.rank == 0 and N == 1 and not args.headline_only:       # PCIe-inclusive rate: single-GPU runs only
            hn = 1 << 30                                          # 1 GiB through the drop-in one-shot beltCTR
            host = np.zeros(hn, dtype=np.uint8)
            hp = ctypes.c_void_p(host.ctypes.data)
            key, iv = bytes(H[128:160]), bytes(H[192:208])
            v, ms = host_api_rate(lambda: eng._check(eng.lib.beltCTR(hp, hp, ctypes.c_size_t(hn), key,
                                                                       ctypes.c_size_t(32), iv), "beltCTR"), 1.0)
            others["beltCTR"]["host_api"] = {"entry": "beltCTR (bee2 drop-in, belt.h:734)", "value": v, "unit": "GiB/s",
                                             "ms_per_call": ms, "sample": "1 GiB, in place",
                                             "note": "host pointers, PCIe both ways inside the call; 1 GPU; not `value`"}
            del host
        if do_cpu:
            others["beltCTR"]["cpu_baseline"] = cpu_baseline("ctr", hc)
        del buf

    # ------------------------------------------------------------------------- verify
    if "verify" in only:
        import goldenlib                      # committed fixtures only; the oracle is not imported here
        G = goldenlib.Golden()
        hs, ss, ps = G.bign_base_arrays()
        nbase = len(hs) // 32
        reps = (1 << 18) // nbase
        # tile the 2048 genuine triples to 2^18 and corrupt a seeded 1/16;
        # verification cost does not depend on the values, so tiling does not flatter it
        Hh = np.tile(np.frombuffer(hs, dtype=np.uint8), reps).reshape(-1, 32).copy()
        Ss = np.tile(np.frombuffer(ss, dtype=np.uint8), reps).reshape(-1, 48).copy()
        Kk = np.tile(np.frombuffer(ps, dtype=np.uint8), reps).reshape(-1, 64).copy()
        n = Hh.shape[0]
        rng = np.random.default_rng(0xB164 + dist.rank)
        bad = rng.choice(n, n // 16, replace=False)
        Ss[bad, rng.integers(0, 48, bad.size)] ^= (1 << rng.integers(0, 8, bad.size)).astype(np.uint8)
        dh, ds, dk = (torch.from_numpy(x.reshape(-1)).cuda() for x in (Hh, Ss, Kk))
        codes = torch.empty(n, dtype=torch.int32, device="cuda")
        kv = max(3, min(K, 10))
        el = timed(dist, kv, 2, lambda: eng.bign128Verify_batch_dev(dh, ds, dk, codes))
        ms_launch = timed.event_ms
        got = codes.cpu().numpy()
        okmask = np.ones(n, dtype=bool)
        okmask[bad] = False
        sane = bool((got[okmask] == 0).all() and (got[bad] == 510).all())
        others["bignVerify"] = {
            "metric": "bign-curve256v1 verifies/s", "value": N * n * kv / el, "unit": "verifies/s", "steps": kv,
            "ms_per_step": el / kv * 1e3, "verdicts_as_expected": sane,
            "config": {"workload": "bignVerify batch: 2^18 signatures per GPU on bign-curve256v1 (BASELINE configs[3]); "
                                   "2048 genuine triples tiled 128x, seeded 1/16 corrupted"},
            "roofline": {"kernels": "bign_prep+main+slow+inv+tail", "bound": "valu-int", "avg_batch_ms": ms_launch,
                         # 32x32+64 multiply-adds per verify: 976 M x 72 + 685 S x 52 + scaled folds; inversions are division steps (no mads)
                         "mads_per_verify": MADS_PER_VERIFY,
                         "achieved": MADS_PER_VERIFY * n / (ms_launch * 1e-3) / 1e12,
                         "peak": MAD_PEAK_T, "unit": "T v_mad_u64_u32 lane-ops/s",
                         "frac": MADS_PER_VERIFY * n / (ms_launch * 1e-3) / 1e12 / MAD_PEAK_T,
                         "note": "integer-multiplier bound; HBM irrelevant (148 B/signature); peak = measured "
                                 "v_mad_u64_u32 micro-benchmark (profiles/r01_valu_rates_ubench.txt); every mad is "
                                 "paired with a half-rate v_addc_co_u32, so 0.5 is the practical ceiling"},
        }
        # MAD_PEAK_T is one micro-benchmark at 2.31 GHz; the verification kernels run at whatever the box gives under THEM (VERDICT r04
        # weak 11): the shader clock beside bign_main_kernel, and the fraction against the multiplier rate at that clock
        try:
            ghz_v, _ = shader_clock_under(lambda: eng.bign128Verify_batch_dev(dh, ds, dk, codes), ms_launch)
        except Exception:
            ghz_v = None
        rv = others["bignVerify"]["roofline"]
        rv["shader_clock_ghz_under_kernels"] = ghz_v
        rv["peak_at_measured_clock"] = MAD_PEAK_T * ghz_v / MAD_PEAK_GHZ if ghz_v else None
        rv["frac_at_measured_clock"] = rv["achieved"] / rv["peak_at_measured_clock"] if ghz_v else None
        if dist.rank == 0:
            # the latency floor: prefixes of the same device-resident batch; up to 2^15 signatures run
            # one per DPP quad, up to 2^16 on 29-bit limbs, above on 32-bit limbs (profiles/r02_verify_small.txt)
            small = {}
            for e in (10, 13, 14, 15, 16, 17):
                m = 1 << e
                pre = (dh[: 32 * m], ds[: 48 * m], dk[: 64 * m], codes[:m])
                for _ in range(6):
                    eng.bign128Verify_batch_dev(*pre)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    eng.bign128Verify_batch_dev(*pre)
                e1.record()
                torch.cuda.synchronize()
                ms_b = e0.elapsed_time(e1) / 20
                small[f"2^{e}"] = {"ms_per_batch": ms_b, "verifies_per_s": m / (ms_b * 1e-3)}
            others["bignVerify"]["batch_size_sweep"] = small
        strong_leg("verify", n, lambda lo, hi: (lambda: eng.bign128Verify_batch_dev(dh[32 * lo: 32 * hi], ds[48 * lo: 48 * hi],
                                                                                     dk[64 * lo: 64 * hi], codes[lo: hi])), kv,
                   t_total_ms=ms_launch if N == 1 else None)
        if dist.rank == 0 and N == 1:       # PCIe-inclusive rate: single-GPU runs only
            hcodes = np.empty(n, dtype=np.uint32)
            prm = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
            from bee2_amd.engine import OID_BELT_HASH_DER
            oid = bytes(OID_BELT_HASH_DER)
            args_h = (ctypes.byref(prm), oid, ctypes.c_size_t(len(oid)), ctypes.c_void_p(Hh.ctypes.data),
                      ctypes.c_void_p(Ss.ctypes.data), ctypes.c_void_p(Kk.ctypes.data), ctypes.c_size_t(n),
                      ctypes.c_void_p(hcodes.ctypes.data))
            v, ms = host_api_rate(lambda: eng._check(eng.lib.bee2hip_bignVerify_batch(*args_h), "bignVerify_batch"), n)
            others["bignVerify"]["host_api"] = {"entry": "bee2hip_bignVerify_batch", "value": v, "unit": "verifies/s",
                                                "ms_per_call": ms, "same_verdicts": bool((hcodes == got.astype(np.uint32)).all()),
                                                "note": "host pointers, PCIe both ways inside the call; 1 GPU; not `value`"}
        if do_cpu:
            others["bignVerify"]["cpu_baseline"] = cpu_baseline("verify", hc)
        # the step `sig vfy` runs before each verification: bign128PubkeyVal over the same keys (tiled 64x more: 1 GiB, beyond the 256 MiB MALL)
        kk = dk.repeat(64)
        nk = kk.numel() // 64
        kcodes = torch.empty(nk, dtype=torch.int32, device="cuda")
        el = timed(dist, kv, 2, lambda: eng.bignPubkeyValL_batch_dev(128, kk, kcodes))
        ach = 68 * nk / (timed.event_ms * 1e-3) / 1e9
        pv_traffic, pv_traffic_src = pmc_traffic("bign_pubkey_val_kernel")
        others["bignPubkeyVal"] = {
            "metric": "bign-curve256v1 public keys validated/s", "value": N * nk * kv / el, "unit": "keys/s", "steps": kv,
            "ms_per_step": el / kv * 1e3, "all_valid": bool((kcodes == 0).all()),
            "config": {"workload": f"bignPubkeyVal batch: {nk} keys per GPU (the 2048 genuine keys tiled)"},
            "roofline": {"kernel": "bign_pubkey_val_kernel<8>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pv_traffic, "traffic_source": pv_traffic_src,
                         "avg_launch_ms": timed.event_ms,
                         "note": "68 B per key (64 in + 4 out); 2 squarings + 1 multiplication per key are ~0.3 ms of VALU work for "
                                 "2^24 keys, i.e. arithmetic and HBM demands are about equal; wall time per launch"}}
        if do_cpu:
            import refgen
            if refgen.have_ref():
                ref = ctypes.CDLL(refgen.REF_SO)
                ref.bign128PubkeyVal.restype = ctypes.c_uint32
                t0, cnt = time.perf_counter(), 0
                while time.perf_counter() - t0 < 1.0:
                    i = cnt % nbase
                    ref.bign128PubkeyVal(ps[64 * i: 64 * i + 64])
                    cnt += 1
                others["bignPubkeyVal"]["cpu_baseline"] = {
                    "value": cnt / (time.perf_counter() - t0), "unit": "keys/s", "cores": 1, "kind": "reference",
                    "sample": "1 s of bign128PubkeyVal calls, one thread"}
        # many signatures of ONE signer (the `sig vfy` batch over a tree of files): the key is a fixed base with a comb
        # table of its own, no doublings left.  2^18 DISTINCT valid signatures made here by the signing entry (untimed), 1/16 damaged.
        from bee2_amd.engine import OID_BELT_HASH_DER as _OID
        n1 = 1 << 18
        g1 = torch.Generator(device="cuda"); g1.manual_seed(0x51D + dist.rank)
        h1 = torch.empty(32 * n1, dtype=torch.uint8, device="cuda"); h1.view(torch.int64).random_(generator=g1)
        d1 = bytes(range(7, 39))[:31] + b"\x21"
        p1 = torch.empty(64, dtype=torch.uint8, device="cuda"); c1 = torch.empty(1, dtype=torch.int32, device="cuda")
        eng.bignPubkeyCalcL_batch_dev(128, torch.from_numpy(np.frombuffer(d1, dtype=np.uint8).copy()).cuda(), p1, c1)
        s1 = torch.empty(48 * n1, dtype=torch.uint8, device="cuda"); cs = torch.empty(n1, dtype=torch.int32, device="cuda")
        eng.bignSign2L_batch_dev(128, _OID, h1, torch.from_numpy(np.frombuffer(d1 * n1, dtype=np.uint8).copy()).cuda(), s1, cs)
        torch.cuda.synchronize()
        pub1 = p1.cpu().numpy().tobytes()
        bad1 = torch.from_numpy(rng.choice(n1, n1 // 16, replace=False)).cuda()
        s1.view(n1, 48)[bad1, 5] ^= 0x10
        codes1 = torch.empty(n1, dtype=torch.int32, device="cuda")
        # untimed: the first call builds and caches the key's 8-bit comb table; a key gets its 16-bit table once 2^19 signatures have
        # been verified under it -- two calls here -- so the timed calls see the steady state of a busy key (16 + 8 additions)
        for _ in range(3):
            eng.bignVerifyL_onekey_batch_dev(128, _OID, h1, s1, pub1, codes1)
        el = timed(dist, kv, 2, lambda: eng.bignVerifyL_onekey_batch_dev(128, _OID, h1, s1, pub1, codes1))
        ms1 = timed.event_ms
        want1 = torch.zeros(n1, dtype=torch.int32, device="cuda"); want1[bad1] = 510
        MADS_ONEKEY = 24 * 732 + 5 * 72          # 16 (u G) + 8 (v Q) mixed additions (8M + 3S) + x_R; inversions are division steps
        others["bignVerify_onekey"] = {
            "metric": "bign-curve256v1 verifies/s, one signer", "value": N * n1 * kv / el, "unit": "verifies/s", "steps": kv,
            "ms_per_step": el / kv * 1e3, "verdicts_as_expected": bool((codes1 == want1).all() and int(cs.abs().sum()) == 0),
            "vs_general_entry": (n1 * kv / el) / (n * kv / (others["bignVerify"]["ms_per_step"] * 1e-3 * kv)),
            "config": {"workload": "bee2hip_bignVerifyL_onekey_batch_dev: 2^18 distinct signatures under ONE public key per GPU "
                                   "(made by the signing entry, 1/16 damaged); the key's comb tables cached (16-bit windows: the state of a key after 2^19 signatures)"},
            "roofline": {"kernels": "bign_onekey + slow + inv + tail", "bound": "valu-int", "avg_batch_ms": ms1,
                         "mads_per_verify": MADS_ONEKEY, "achieved": MADS_ONEKEY * n1 / (ms1 * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                         "unit": "T v_mad_u64_u32 lane-ops/s", "frac": MADS_ONEKEY * n1 / (ms1 * 1e-3) / 1e12 / MAD_PEAK_T}}
        if dist.rank == 0:
            # the latency floor: prefixes of the same batch (up to 2^16 signatures four lanes share a signature)
            small1 = {}
            for e in (10, 14, 15, 16, 17):
                m = 1 << e
                pre = (h1[: 32 * m], s1[: 48 * m], pub1, codes1[:m])
                for _ in range(4):
                    eng.bignVerifyL_onekey_batch_dev(128, _OID, *pre)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    eng.bignVerifyL_onekey_batch_dev(128, _OID, *pre)
                e1.record()
                torch.cuda.synchronize()
                ms_b = e0.elapsed_time(e1) / 20
                small1[f"2^{e}"] = {"ms_per_batch": ms_b, "verifies_per_s": m / (ms_b * 1e-3)}
            others["bignVerify_onekey"]["batch_size_sweep"] = small1
        # ... and of a FEW signers: the population SURVEY 8d describes (64 key pairs), every signature distinct, 1/16 damaged
        nk = 64
        dks = [bytes(((k * 37 + i * 11 + 5) & 255) for i in range(31)) + b"\x21" for k in range(nk)]
        pk = torch.empty(64 * nk, dtype=torch.uint8, device="cuda"); ck = torch.empty(nk, dtype=torch.int32, device="cuda")
        eng.bignPubkeyCalcL_batch_dev(128, torch.from_numpy(np.frombuffer(b"".join(dks), dtype=np.uint8).copy()).cuda(), pk, ck)
        kidx = torch.from_numpy(rng.integers(0, nk, n1).astype(np.int32)).cuda()
        dall = torch.from_numpy(np.frombuffer(b"".join(dks), dtype=np.uint8).copy()).cuda().view(nk, 32)[kidx.long()].reshape(-1).contiguous()
        eng.bignSign2L_batch_dev(128, _OID, h1, dall, s1, cs)
        torch.cuda.synchronize()
        s1.view(n1, 48)[bad1, 5] ^= 0x10
        pubs_k = pk.cpu().numpy().tobytes()
        eng.bignVerifyL_keyed_batch_dev(128, _OID, h1, s1, pubs_k, kidx, codes1)       # untimed: the 64 tables are built and cached
        el = timed(dist, kv, 2, lambda: eng.bignVerifyL_keyed_batch_dev(128, _OID, h1, s1, pubs_k, kidx, codes1))
        msk = timed.event_ms
        MADS_KEYED = 32 * 732 + 5 * 72           # 16 (u G) + 16 (v Q, 8-bit windows) mixed additions + x_R
        others["bignVerify_keyed"] = {
            "metric": "bign-curve256v1 verifies/s, 64 signers", "value": N * n1 * kv / el, "unit": "verifies/s", "steps": kv,
            "ms_per_step": el / kv * 1e3, "verdicts_as_expected": bool((codes1 == want1).all() and int(cs.abs().sum()) == 0 and int(ck.abs().sum()) == 0),
            "vs_general_entry": (n1 * kv / el) / (n / (others["bignVerify"]["ms_per_step"] * 1e-3)),
            "config": {"workload": "bee2hip_bignVerifyL_keyed_batch_dev: 2^18 distinct signatures of 64 signers per GPU (random signer per "
                                   "signature, 1/16 damaged); the signers' 8-bit comb tables cached (17 MiB)"},
            "roofline": {"kernels": "bign_onekey<keyed> + slow + inv + tail", "bound": "valu-int", "avg_batch_ms": msk,
                         "mads_per_verify": MADS_KEYED, "achieved": MADS_KEYED * n1 / (msk * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                         "unit": "T v_mad_u64_u32 lane-ops/s", "frac": MADS_KEYED * n1 / (msk * 1e-3) / 1e12 / MAD_PEAK_T}}
        del h1, s1, cs, codes1, want1, kidx, dall, pk
        del dh, ds, dk, codes, kk, kcodes

    # ------------------------------------------------- 8f-4: the 384- and 512-bit curves
    if "verify" in only and "bign_big" in G.__dict__:
        from bee2_amd.engine import LEVEL_OID
        for l in (192, 256):
            base = G.bign_big[str(l)]["base"]
            reps_l = (1 << 18) // len(base)
            hs_l = b"".join(bytes.fromhex(t["hash"]) for t in base) * reps_l
            ss_l = b"".join(bytes.fromhex(t["sig"]) for t in base) * reps_l
            ps_l = b"".join(bytes.fromhex(t["pubkey"]) for t in base) * reps_l
            nl = len(base) * reps_l
            th, ts, tp = (torch.from_numpy(np.frombuffer(x, dtype=np.uint8).copy()).cuda() for x in (hs_l, ss_l, ps_l))
            tc = torch.empty(nl, dtype=torch.int32, device="cuda")
            kl = max(2, min(K, 5))
            el = timed(dist, kl, 1, lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], th, ts, tp, tc))
            others[f"bignVerify_l{l}"] = {
                "metric": f"bign-curve{2 * l}v1 verifies/s", "value": N * nl * kl / el, "unit": "verifies/s",
                "steps": kl, "ms_per_step": el / kl * 1e3, "all_valid": bool((tc == 0).all()),
                "config": {"workload": f"{nl} signatures per GPU on the {2 * l}-bit curve (SURVEY 8f-4; 2^18 as configs[3]: 2^16 leaves one wavefront per SIMD), "
                                       f"{len(base)} genuine triples tiled"}}
            if dist.rank == 0:
                # small batches (prefixes of the same tiling): the quad kernel up to 2^14 signatures, the r01 kernels above
                small = {}
                no_l = l // 4
                for e in (10, 13, 14, 15):
                    m = 1 << e
                    args = (th[: no_l * m], ts[: (no_l + no_l // 2) * m], tp[: 2 * no_l * m], tc[:m])
                    for _ in range(2):
                        eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], *args)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], *args)
                    e1.record()
                    torch.cuda.synchronize()
                    ms_b = e0.elapsed_time(e1) / 5
                    small[f"2^{e}"] = {"ms_per_batch": ms_b, "verifies_per_s": m / (ms_b * 1e-3)}
                others[f"bignVerify_l{l}"]["batch_size_sweep"] = small
            if do_cpu:
                import refgen
                if refgen.have_ref():
                    ref = ctypes.CDLL(refgen.REF_SO)
                    f = getattr(ref, f"bign{l}Verify")
                    f.restype = ctypes.c_uint32
                    no = l // 4
                    t0, cnt = time.perf_counter(), 0
                    while time.perf_counter() - t0 < 1.5:
                        i = cnt % len(base)
                        f(hs_l[no * i: no * i + no], ss_l[(no + no // 2) * i: (no + no // 2) * (i + 1)],
                          ps_l[2 * no * i: 2 * no * (i + 1)])
                        cnt += 1
                    others[f"bignVerify_l{l}"]["cpu_baseline"] = {
                        "value": cnt / (time.perf_counter() - t0), "unit": "verifies/s", "cores": 1,
                        "kind": "reference", "sample": "1.5 s of bign%dVerify calls, one thread" % l}
            del th, ts, tp, tc

    # ---------------------------------------------- 8f-4 tail: key generation / signing (constant-time kernels)
    if "sign" in only:
        from bee2_amd.engine import LEVEL_OID
        l, no, sg = 128, 32, 48
        n = 1 << 18
        privs = torch.empty(no * n, dtype=torch.uint8, device="cuda")
        fill_seeded(privs, 0x5164 + dist.rank)
        privs.view(-1, no)[:, no - 1] &= 0x7F                       # d < 2^255 < q: every key valid
        privs.view(-1, no)[:, 0] |= 1
        hsh = torch.empty(no * n, dtype=torch.uint8, device="cuda")
        fill_seeded(hsh, 0x5165 + dist.rank)
        sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
        pubs = torch.empty(2 * no * n, dtype=torch.uint8, device="cuda")
        sc = torch.empty(n, dtype=torch.int32, device="cuda")
        ks = max(3, min(K, 10))
        el = timed(dist, ks, 2, lambda: eng.bignSign2L_batch_dev(l, LEVEL_OID[l], hsh, privs, sigs, sc))
        ms_sign = timed.event_ms
        el_k = timed(dist, ks, 2, lambda: eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, sc))
        ms_calc = timed.event_ms
        vc = torch.empty(n, dtype=torch.int32, device="cuda")
        eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], hsh, sigs, pubs, vc)        # not timed: every signature must verify
        torch.cuda.synchronize()
        # 32x32+64 multiply-adds per signature: 33 signed 8-bit windows (round 4; 43 of 6 bits in round 3, 64 of 4 bits before) x 11
        # multiplications x (64 + 8) -- see below for the count of the Jacobian form -- and the affine coordinates; the inversion (fixed-count division steps,
        # fe_inv_safegcd<N, true>), the belt work (16 block encryptions) and the table scan have none
        # (8 M + 3 S per Jacobian mixed addition: 8 x 72 + 3 x 52 multiply-adds with the reductions; 4 M + 1 S for x, y)
        mads = 33 * (8 * 72 + 3 * 52) + 2 * 72 + 52     # round 4: 33 signed 8-bit windows looked up in LDS (bign_mulbase_lds_kernel); x_R only
        others["bignSign2"] = {
            "metric": "bign-curve256v1 deterministic signatures/s", "value": N * n * ks / el, "unit": "signatures/s", "steps": ks,
            "ms_per_step": el / ks * 1e3, "all_verify": bool((vc == 0).all() and (sc == 0).all()),
            "config": {"workload": f"bignSign2 batch: {n} (hash, private key) pairs per GPU on bign-curve256v1, no additional input; "
                                   "constant-time kernels (nonce by belt-hash + belt-wbl, k G on signed 8-bit windows whose entry is looked up in bank-private LDS copies of the row, "
                                   "masked Jacobian mixed additions, inversion by a fixed number of division steps); every signature verified afterwards (untimed)"},
            "roofline": {"kernels": "bign_sign_nonce + bign_mulbase_lds (one lane per signature, window entries looked up in LDS) + bign_sign_tail", "bound": "valu-int", "avg_batch_ms": ms_sign,
                         "mads_per_signature": mads, "achieved": mads * n / (ms_sign * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                         "unit": "T v_mad_u64_u32 lane-ops/s", "frac": mads * n / (ms_sign * 1e-3) / 1e12 / MAD_PEAK_T,
                         "note": "same multiplier formulation as verification (each mad paired with a half-rate addc: 0.5 is the ceiling)"},
            "pubkey_calc": {"value": N * n * ks / el_k, "unit": "keys/s", "ms_per_step": el_k / ks * 1e3, "avg_batch_ms": ms_calc},
        }
        if dist.rank == 0:
            # prefixes of the same device-resident batch: k G runs on 64 / 16 / 4 lanes per signature up to 2^10 / 2^13 / 2^16
            # signatures, one lane above (profiles/r03_sign_coop.txt); wall clock around launch + synchronise, best of 5
            small = {}
            for e in (0, 10, 12, 14, 15, 16):
                m = 1 << e
                best = 1e9
                for _ in range(6):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.bignSign2L_batch_dev(l, LEVEL_OID[l], hsh[: no * m], privs[: no * m], sigs[: sg * m], sc[:m])
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                small[f"2^{e}"] = {"ms_per_batch": best * 1e3, "signatures_per_s": m / best}
            others["bignSign2"]["batch_size_sweep"] = small
        if do_cpu:
            import refgen
            if refgen.have_ref():
                ref = ctypes.CDLL(refgen.REF_SO)
                ref.bign128Sign2.restype = ctypes.c_uint32
                hh, pp = hsh[: no * 512].cpu().numpy().tobytes(), privs[: no * 512].cpu().numpy().tobytes()
                out = ctypes.create_string_buffer(sg)
                t0, cnt = time.perf_counter(), 0
                while time.perf_counter() - t0 < 3.0:
                    i = cnt % 512
                    ref.bign128Sign2(out, hh[no * i: no * i + no], pp[no * i: no * i + no], None, ctypes.c_size_t(0))
                    cnt += 1
                others["bignSign2"]["cpu_baseline"] = {
                    "value": cnt / (time.perf_counter() - t0), "unit": "signatures/s", "cores": 1, "kind": "reference",
                    "sample": "3 s of bign128Sign2 calls on 512 of the same (hash, key) pairs, one thread"}
        del privs, hsh, sigs, pubs, sc, vc

    # ------------------------------------------- single-call latency of the drop-in entry points (host pointers)
    if "latency" in only and dist.rank == 0 and N == 1:
        lat = {}

        def us_per_call(fn, reps):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t0) / reps * 1e6
        blk = ctypes.create_string_buffer(192)
        st_ctr = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
        eng.lib.beltCTRStart(st_ctr, H[128:160], ctypes.c_size_t(32), H[192:208])
        b16 = ctypes.create_string_buffer(16)
        b64k = ctypes.create_string_buffer(1 << 16)
        import goldenlib
        Gk = goldenlib.Golden()
        h0, s0, p0 = Gk.bign_base[0]
        d0 = bytes(range(1, 33))
        sg0 = ctypes.create_string_buffer(48)

        def measure(fast):
            lat = {}
            lat["bashF (192 B)"] = us_per_call(lambda: eng.lib.bashF(blk, None), 20000 if fast else 200)
            lat["beltCTRStepE (16 B)"] = us_per_call(lambda: eng.lib.beltCTRStepE(b16, ctypes.c_size_t(16), st_ctr), 20000 if fast else 200)
            lat["beltCTRStepE (64 KiB)"] = us_per_call(lambda: eng.lib.beltCTRStepE(b64k, ctypes.c_size_t(1 << 16), st_ctr), 100)
            lat["bign128Verify"] = us_per_call(lambda: eng.lib.bign128Verify(h0, s0, p0), 1000 if fast else 50)
            lat["bign128PubkeyVal"] = us_per_call(lambda: eng.lib.bign128PubkeyVal(p0), 20000 if fast else 100)
            lat["bign128Sign2"] = us_per_call(lambda: eng.lib.bign128Sign2(sg0, h0, d0, None, ctypes.c_size_t(0)), 50)
            lat["beltHash (1 KiB)"] = us_per_call(lambda: eng.lib.beltHash(ctypes.create_string_buffer(32), bytes(1024), ctypes.c_size_t(1024)), 2000 if fast else 100)
            return lat
        lat = measure(True)                                        # the default: small calls on the host path, by size
        eng.lib.bee2hip_path_policy(1)                             # as BEE2HIP_FORCE=gpu: every primitive in a kernel (rounds 1-2)
        lat_gpu = measure(False)
        eng.lib.bee2hip_path_policy(0)
        entry = {"unit": "us per call", "dropin": lat, "dropin_forced_gpu": lat_gpu,
                 "note": "dropin = the library as a caller gets it: single primitives, block-parallel modes under 8 KiB per call, "
                         "one-message serial chains and ONE signature verification / public-key validation run on the host path "
                         "(bee2_amd/csrc/host_small.hpp, host_bign.hpp; nothing with a private key does), everything else is H2D + "
                         "launch(es) + D2H on the NULL stream; dropin_forced_gpu = BEE2HIP_FORCE=gpu (every primitive in a kernel); "
                         "the batch entry points are the fast path (INTEGRATION.md gives the crossover sizes)"}
        if do_cpu:
            import refgen
            if refgen.have_ref():
                ref = ctypes.CDLL(refgen.REF_SO)
                cpu = {}
                cpu["bashF (192 B)"] = us_per_call(lambda: ref.bashF(blk, None), 20000)
                rst = ctypes.create_string_buffer(ref.beltCTR_keep())
                ref.beltCTRStart(rst, H[128:160], ctypes.c_size_t(32), H[192:208])
                cpu["beltCTRStepE (16 B)"] = us_per_call(lambda: ref.beltCTRStepE(b16, ctypes.c_size_t(16), rst), 20000)
                cpu["beltCTRStepE (64 KiB)"] = us_per_call(lambda: ref.beltCTRStepE(b64k, ctypes.c_size_t(1 << 16), rst), 500)
                cpu["bign128Verify"] = us_per_call(lambda: ref.bign128Verify(h0, s0, p0), 300)
                cpu["bign128PubkeyVal"] = us_per_call(lambda: ref.bign128PubkeyVal(p0), 5000)
                cpu["bign128Sign2"] = us_per_call(lambda: ref.bign128Sign2(sg0, h0, d0, None, ctypes.c_size_t(0)), 300)
                cpu["beltHash (1 KiB)"] = us_per_call(lambda: ref.beltHash(ctypes.create_string_buffer(32), bytes(1024), ctypes.c_size_t(1024)), 2000)
                entry["cpu_reference"] = cpu
        others["single_call_latency_us"] = entry

    # -------------------------------------------------------------------------- mixed
    if "mixed" in only:
        n, ml = 1 << 21, 4096                                      # the weak leg: 2^24 / 8 messages per GPU
        # the FIXED job of configs[4]: 2^24 x 4 KiB = 64 GiB, resident on this card when it fits (288 GB: four times over); the weak
        # leg runs over its first 2^21 messages, the whole job is `bash512_beltMAC_2p24` and the total of the strong split
        n_all = STRONG_TOTALS["mixed"]
        free, _ = torch.cuda.mem_get_info()
        while n_all * (ml + 72) + (2 << 30) > free and n_all > 1024:
            n_all //= 2
        n_all = int(round(-dist.max(-float(n_all))))               # the same job on every rank (the strong legs are collective): the smallest
        n = min(n, n_all)
        msgs = torch.empty(n_all * ml, dtype=torch.uint8, device="cuda")
        fill_seeded(msgs, 0x4D1C + dist.rank)
        dig_all = torch.empty(n_all * 64, dtype=torch.uint8, device="cuda")
        tag_all = torch.empty(n_all * 8, dtype=torch.uint8, device="cuda")
        dig, tag = dig_all[: n * 64], tag_all[: n * 8]
        km = max(2, min(K, 5))
        mixed_unit = lambda lo, hi: (lambda: eng.bashHash_beltMAC_batch_dev(msgs[ml * lo: ml * hi], ml, 256, H[128:160],  # noqa: E731
                                                                             dig_all[64 * lo: 64 * hi], tag_all[8 * lo: 8 * hi]))
        el = timed(dist, km, 1, mixed_unit(0, n))
        ms_mixed = timed.event_ms
        # the two parts on this GPU in this run: taken from the bashF / beltCTR legs above, or (--only mixed) short legs here
        src = "the bashF and beltCTR legs of this run (kernel time, per GPU)"
        if "bashF_perms_per_s" not in rates or "belt_blocks_per_s" not in rates:
            src = "short bashF (2^20 states) and beltCTR (1 GiB) legs run for this roofline (kernel time, per GPU)"
            if "bashF_perms_per_s" not in rates:
                stp = msgs[: 192 << 20]
                timed(dist, 20, 3, lambda: eng.bashF_batch_dev(stp))
                rates["bashF_perms_per_s"] = (1 << 20) / (timed.event_ms * 1e-3)
            if "belt_blocks_per_s" not in rates:
                cb_ = msgs[: 1 << 30] if msgs.numel() >= (1 << 30) else msgs
                timed(dist, 3, 1, lambda: eng.beltCTR_blocks_dev(cb_, kw, c0, 0))
                rates["belt_blocks_per_s"] = (cb_.numel() // 16) / (timed.event_ms * 1e-3)
            fill_seeded(msgs[: 1 << 30] if msgs.numel() >= (1 << 30) else msgs, 0x4D1C + dist.rank)   # (the short legs ran in place over the first messages)
        others["bash512_beltMAC"] = {
            "metric": "bash512+beltMAC messages/s", "value": N * n * km / el, "unit": "messages/s", "steps": km,
            "ms_per_step": el / km * 1e3, "GiB_per_s": N * n * ml * km / el / 2 ** 30,
            "config": {"workload": f"bash512 + beltMAC over {n} x 4 KiB messages per GPU (BASELINE configs[4] share of one GPU)"},
            "roofline": dict(mixed_roofline(n / (ms_mixed * 1e-3), rates["bashF_perms_per_s"], rates["belt_blocks_per_s"], src),
                             kernel="hash_mac_fused_kernel<8, true, true, BeltTabWide>", avg_launch_ms=ms_mixed,
                             hbm_frac=(4096 + 72) * n / (ms_mixed * 1e-3) / 1e9 / HBM_PEAK_GBS),
        }
        if n_all > n:
            ks_ = max(2, min(K, 3))
            if N == 1:
                t_all = event_ms(mixed_unit(0, n_all), ks_, warmup=1)
                others["bash512_beltMAC_2p24"] = {
                    "metric": "bash512+beltMAC messages/s, the whole configs[4] job on ONE GPU", "value": n_all / (t_all * 1e-3),
                    "unit": "messages/s", "steps": ks_, "ms_per_step": t_all, "GiB_per_s": n_all * ml / (t_all * 1e-3) / 2 ** 30,
                    "config": {"workload": f"bash512 + beltMAC over {n_all} x 4 KiB messages = {n_all * ml / 2**30:.0f} GiB resident on one GPU, "
                                           "ONE bee2hip_bashHash_beltMAC_batch_dev call per step (BASELINE configs[4], the N = 1 point)"}}
                strong_leg("mixed", n_all, mixed_unit, ks_, t_total_ms=t_all)
            else:
                strong_leg("mixed", n_all, mixed_unit, ks_)
        if do_cpu:
            others["bash512_beltMAC"]["cpu_baseline"] = cpu_baseline("mixed", hc)
        del msgs, dig, tag, dig_all, tag_all

    # ------------------------------------------------- 8f-3: ragged hash batches (bsum front-end)
    if "ragged" in only:
        nm = 1 << 16
        rng = np.random.default_rng(0x4D1C + dist.rank)
        lens = np.floor(2.0 ** (18.0 * rng.random(nm))).astype(np.int64) - 1      # log-uniform in [0, 256 KiB)
        entry = {"metric": "ragged hash GiB/s", "unit": "GiB/s",
                 "config": {"workload": f"{nm} messages per GPU, lengths log-uniform in [0, 256 KiB) (seed 0x4D1C), "
                                        f"{int(lens.sum()) / 2**30:.2f} GiB, packed back to back (SURVEY 8f-3)"}}
        kr = max(2, min(K, 5))
        offs = np.zeros(nm + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        total = int(offs[-1])
        data = torch.empty(max(total, 8) // 8 * 8 + 8, dtype=torch.uint8, device="cuda")
        fill_seeded(data, 0x4D1C + dist.rank)
        doff = torch.from_numpy(offs).cuda()
        # "caller_order": no order passed, the library buckets the lengths on the device (powers of two, longest
        # bucket first); "longest_first": an exact descending sort passed in (what the host entry point does itself)
        dord = torch.from_numpy(np.argsort(-lens, kind="stable").astype(np.int32)).cuda()
        for name, alg, dl in (("belt_hash", 0, 32), ("bash256", 128, 32)):
            dig = torch.empty(nm * dl, dtype=torch.uint8, device="cuda")
            for order, o in (("caller_order", None), ("longest_first", dord)):
                el = timed(dist, kr, 1, lambda: eng.hash_ragged_dev(alg, data, doff, dig, nm, order=o))
                entry[f"{name}_{order}"] = N * total * kr / el / 2 ** 30
            if name == "belt_hash":                               # bsum's default algorithm (bsum.c:392-394)
                entry["value"] = entry["belt_hash_longest_first"]
                entry["steps"] = kr
                entry["ms_per_step"] = el / kr * 1e3
                # what bounds it: belt-hash of ONE message is a serial chain (two dependent encryptions per 32 bytes), so the batch
                # cannot finish before its longest message does -- that message alone, beside the batch
                big = int(np.argmax(lens))
                one_off = torch.from_numpy(np.array([0, int(lens[big])], dtype=np.int64)).cuda()
                one = data[int(offs[big]) // 8 * 8:]                   # (an aligned view; the chain's time does not depend on the bytes)
                one_dig = torch.empty(32, dtype=torch.uint8, device="cuda")
                el1 = timed(dist, kr, 1, lambda: eng.hash_ragged_dev(0, one, one_off, one_dig, 1))
                entry["roofline"] = {"bound": "latency of one serial chain", "longest_message_bytes": int(lens[big]),
                                     "longest_chain_alone_ms": el1 / kr * 1e3,
                                     "frac": (el1 / kr) / (el / kr),
                                     "note": "frac = the longest message hashed ALONE / the whole batch: a lone wavefront gets one "
                                             "issue slot per 4 cycles and ~56 cycles per LDS round trip, ~3600 cycles per belt "
                                             "encryption (profiles/r04_long_hash_ab.txt, tools/ubench/lone_chain.hip)"}
                del one, one_off, one_dig
            if do_cpu:
                import refgen
                if refgen.have_ref():
                    from concurrent.futures import ThreadPoolExecutor
                    ref = ctypes.CDLL(refgen.REF_SO)
                    sub = min(nm, 8192)                                # bounded sample: the first 8192 messages
                    host = data[: int(offs[sub])].cpu().numpy()
                    base = host.ctypes.data
                    outs = np.empty((sub, dl), dtype=np.uint8)

                    def work(r, name=name, outs=outs, base=base):
                        for i in r:
                            src, cnt = ctypes.c_void_p(base + int(offs[i])), ctypes.c_size_t(int(lens[i]))
                            dst = ctypes.c_void_p(outs[i].ctypes.data)
                            if name == "belt_hash":
                                ref.beltHash(dst, src, cnt)
                            else:
                                ref.bashHash(dst, ctypes.c_size_t(128), src, cnt)
                    nthr = min(cores, 64)
                    parts = [range(t, sub, nthr) for t in range(nthr)]
                    t0 = time.perf_counter()
                    with ThreadPoolExecutor(nthr) as ex:
                        list(ex.map(work, parts))
                    dt = time.perf_counter() - t0
                    gpu = dig[: sub * dl].cpu().numpy().reshape(sub, dl)
                    entry.setdefault("cpu_baseline", {"kind": "reference", "cores": nthr, "unit": "GiB/s",
                                                      "sample": f"first {sub} messages, {nthr} threads calling "
                                                                "beltHash / bashHash of the reference"})
                    entry["cpu_baseline"][name] = int(offs[sub]) / dt / 2 ** 30
                    entry["cpu_baseline"][f"{name}_digests_equal"] = bool((gpu == outs).all())
                    if name == "belt_hash":
                        entry["cpu_baseline"]["value"] = entry["cpu_baseline"][name]
            del dig
        del data, doff, dord
        # the many-small-files shape of bsum: 2^18 messages of 1000 bytes (packed, so three in four start misaligned)
        nu, lu = 1 << 18, 1000
        udata = torch.empty(nu * lu + 16, dtype=torch.uint8, device="cuda")
        fill_seeded(udata[: (nu * lu) // 8 * 8], 0x4D1C + 7 + dist.rank)
        uoff = torch.arange(nu + 1, dtype=torch.int64, device="cuda") * lu
        udig = torch.empty(nu * 32, dtype=torch.uint8, device="cuda")
        for name, alg in (("belt_hash", 0), ("bash256", 128)):
            el = timed(dist, kr, 1, lambda: eng.hash_ragged_dev(alg, udata, uoff, udig, nu))
            entry[f"{name}_uniform_1000B"] = N * nu * lu * kr / el / 2 ** 30
        del udata, uoff, udig
        others["hash_ragged"] = entry

    # ------------------------------------------------- 8f-2: belt-dwp (CTR + polynomial MAC)
    if "dwp" in only:
        nbytes = 4 << 30
        free, _ = torch.cuda.mem_get_info()
        if free < nbytes + (1 << 30):
            nbytes = (int(free * 0.5) // (1 << 20)) << 20
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        fill_seeded(buf, 0xBE17 + dist.rank)
        dkw, dc0, dr, dt0 = eng.beltDWPStart(H[128:160], H[192:208])
        tout = torch.zeros(16, dtype=torch.uint8, device="cuda")
        kd = max(3, min(K, 10))

        def wrap_dev():                       # beltDWPWrap on resident data: encrypt in place, authenticate the ciphertext
            eng.beltCTR_blocks_dev(buf, dkw, dc0, 0)
            eng.beltDWP_absorb_dev(buf, nbytes, dr, dt0, tout)
        el = timed(dist, kd, 1, wrap_dev)
        el_mac = timed(dist, kd, 1, lambda: eng.beltDWP_absorb_dev(buf, nbytes, dr, dt0, tout))
        ckw, cs0, ct0 = eng.beltCHEStart(H[128:160], H[192:208])

        def che_wrap_dev():                   # beltCHEWrap on resident data (r = s0 for belt-che)
            eng.beltCHE_blocks_dev(buf, buf, ckw, cs0, 0)
            eng.beltDWP_absorb_dev(buf, nbytes, cs0, ct0, tout)
        el_che = timed(dist, kd, 1, che_wrap_dev)
        entry = {"metric": "belt-dwp wrap GiB/s", "unit": "GiB/s", "value": N * nbytes * kd / el / 2 ** 30, "steps": kd,
                 "ms_per_step": el / kd * 1e3, "mac_only": N * nbytes * kd / el_mac / 2 ** 30,
                 "che_wrap": N * nbytes * kd / el_che / 2 ** 30,
                 "config": {"workload": f"{nbytes / 2**30:.0f} GiB message per GPU, device resident: beltCTR_blocks_dev + "
                                        "beltDWP_absorb_dev over the ciphertext (each rank its own message; SURVEY 8f-2)"}}
        if do_cpu:
            import refgen
            if refgen.have_ref():
                from concurrent.futures import ThreadPoolExecutor
                ref = ctypes.CDLL(refgen.REF_SO)
                nthr = min(cores, 64)
                per = 4 << 20                                      # 4 MiB message per thread
                hb = np.ones(nthr * per, dtype=np.uint8)
                macs = np.zeros((nthr, 8), dtype=np.uint8)
                key, iv = bytes(H[128:160]), bytes(H[192:208])

                def work(t):
                    p = ctypes.c_void_p(hb.ctypes.data + t * per)
                    ref.beltDWPWrap(p, ctypes.c_void_p(macs[t].ctypes.data), p, ctypes.c_size_t(per), None,
                                    ctypes.c_size_t(0), key, ctypes.c_size_t(32), iv)
                t0 = time.perf_counter()
                with ThreadPoolExecutor(nthr) as ex:
                    list(ex.map(work, range(nthr)))
                dt = time.perf_counter() - t0
                # same message through the GPU drop-in: the tags must agree
                one = np.ones(per, dtype=np.uint8)
                code, _, gmac = eng.dwp_wrap(one.tobytes(), b"", key, iv)
                entry["cpu_baseline"] = {"kind": "reference", "cores": nthr, "unit": "GiB/s", "value": nthr * per / dt / 2 ** 30,
                                         "sample": f"{nthr} threads, one 4 MiB beltDWPWrap each",
                                         "mac_equal": bool(code == 0 and gmac == macs[0].tobytes())}
                cm = (ctypes.c_ubyte * 8)()
                cb = np.ones(per, dtype=np.uint8)
                t0 = time.perf_counter()
                ref.beltCHEWrap(ctypes.c_void_p(cb.ctypes.data), cm, ctypes.c_void_p(cb.ctypes.data), ctypes.c_size_t(per),
                                None, ctypes.c_size_t(0), key, ctypes.c_size_t(32), iv)
                dt1 = time.perf_counter() - t0
                code, _, gmac = eng.dwp_wrap(one.tobytes(), b"", key, iv, "CHE")
                entry["cpu_baseline"]["che_wrap_single_thread"] = per / dt1 / 2 ** 30
                entry["cpu_baseline"]["che_mac_equal"] = bool(code == 0 and gmac == bytes(cm))
        others["belt_dwp"] = entry
        del buf, tout

    # ------------------------------------------------- 8f-1: ECB / CBC-decrypt bulk modes
    if "modes" in only:
        nbytes = 4 << 30
        free, _ = torch.cuda.mem_get_info()
        if free < 2 * nbytes + (1 << 30):
            nbytes = (int(free * 0.3) // (1 << 20)) << 20
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        fill_seeded(src, 0xBE17 + dist.rank)
        dst = torch.empty_like(src)
        km = max(3, min(K, 10))
        entry = {"metric": "belt ECB/CBC/BDE/SDE bulk GiB/s", "unit": "GiB/s", "steps": km,
                 "config": {"workload": f"{nbytes / 2**30:.0f} GiB of full blocks per GPU, one key (SURVEY 8f-1)"}}
        for name, mode in (("ecb_encr", 0), ("ecb_decr", 1), ("cbc_decr", 2)):
            el = timed(dist, km, 1, lambda: eng.beltModes_blocks_dev(mode, src, dst, kw, c0))
            entry[name] = N * nbytes * km / el / 2 ** 30
        # belt-bde: rank r owns blocks [r*nb, (r+1)*nb) of one stream (first_block), like CTR
        bkw, bs0 = eng.beltBDEStart(H[128:160], H[192:208])
        for name, decr in (("bde_encr", 0), ("bde_decr", 1)):
            el = timed(dist, km, 1, lambda: eng.beltBDE_blocks_dev(decr, src, dst, bkw, bs0,
                                                                   first_block=dist.rank * (nbytes // 16)))
            entry[name] = N * nbytes * km / el / 2 ** 30
        # belt-sde: independent sectors, one lane each (a sector is a serial chain of 2 E per block)
        for sb in (512, 4096):
            ns = (1 << 20) if sb == 512 else (1 << 19)             # 512 MiB / 2 GiB of sectors: >= 8 wavefronts per SIMD
            ns = min(ns, nbytes // sb, dst.numel() // 16)
            sec = src[: ns * sb]
            sivs = dst[: 16 * ns]
            fill_seeded(sivs, 0x5DE + dist.rank)
            for name, decr in ((f"sde_encr_{sb}", 0), (f"sde_decr_{sb}", 1)):
                el = timed(dist, km, 1, lambda: eng.beltSDE_sectors_dev(decr, sec, sb, kw, sivs))
                entry[name] = N * ns * sb * km / el / 2 ** 30
        # CBC encryption is a serial chain per message: a batch of independent 512-byte messages, one lane each
        nm, mb = 1 << 20, 512
        cmsgs = src[: nm * mb]
        civs = dst[: 16 * nm]
        fill_seeded(civs, 0xCBC + dist.rank)
        el = timed(dist, km, 1, lambda: eng.beltCBCEncr_batch_dev(cmsgs, mb // 16, kw, civs))
        entry["cbc_encr_batch_512"] = N * nm * mb * km / el / 2 ** 30
        entry["value"] = entry["ecb_encr"]
        entry["ms_per_step"] = nbytes / 2 ** 30 / entry["ecb_encr"] * 1e3 * N
        if do_cpu:
            import orclib
            import refgen
            if refgen.have_ref():
                orc = orclib.load()
                ref = ctypes.CDLL(refgen.REF_SO)
                hb = np.zeros(64 << 20, dtype=np.uint8)
                ho = np.empty_like(hb)
                cpu = {"cores": cores, "kind": "reference", "unit": "GiB/s",
                       "sample": "64 MiB of full blocks, threads over disjoint slices (belt-sde: each slice one sector)"}
                for name, fn, iv in (("ecb_encr", "beltECBEncr", None), ("ecb_decr", "beltECBDecr", None),
                                     ("cbc_decr", "beltCBCDecr", H[192:208]), ("bde_encr", "beltBDEEncr", H[192:208]),
                                     ("bde_decr", "beltBDEDecr", H[192:208]), ("sde_encr", "beltSDEEncr", H[192:208]),
                                     ("sde_decr", "beltSDEDecr", H[192:208])):
                    fp = ctypes.cast(getattr(ref, fn), ctypes.c_void_p)
                    t0, reps = time.perf_counter(), 0
                    while time.perf_counter() - t0 < 1.5:
                        orc.lib.orc_drive_ref_mode(fp, ctypes.c_void_p(hb.ctypes.data), ctypes.c_void_p(ho.ctypes.data),
                                                   ctypes.c_size_t(hb.nbytes // 16), H[128:160], ctypes.c_size_t(32),
                                                   iv, cores)
                        reps += 1
                    cpu[name] = reps * hb.nbytes / 2 ** 30 / (time.perf_counter() - t0)
                cpu["value"] = cpu["ecb_encr"]
                entry["cpu_baseline"] = cpu
        others["belt_modes"] = entry
        del src, dst

    if not result:                        # --only without bashF: promote the first other metric
        k0 = next(iter(others))
        o = others.pop(k0)
        result = {"metric": o.get("metric", k0), "value": o.get("value"), "unit": o.get("unit"), "n_gpus": N, "steps": o.get("steps", K),
                  "warmup": W, "ms_per_step": o.get("ms_per_step"), "higher_is_better": True, "scaling": "weak",
                  "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": o.get("config", {"workload": k0})}
        for k, v in o.items():            # keep the workload's own fields (roofline, cpu_baseline, extras)
            result.setdefault(k, v)
    # The driver's record keeps `roofline` and `cpu_baseline` as FLAT objects of about two dozen scalars, in order, and cuts
    # strings at 120 characters (BENCH_r03.json lost bignVerify_frac that way): so every fraction and rate FIRST, in a fixed
    # order, then the N > 1 diagnostics, then the rest; prose and nested objects live in `others`.
    # Per-GPU figures for the fractions (value / N), whole-job figures for the rates.
    rf0 = result.get("roofline") or {}
    flat = {k: rf0.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    flat["frac_2p22"] = rf0.get("frac_2p22")
    cb = result.get("cpu_baseline")
    if cb is None:
        cb = result["cpu_baseline"] = {"value": None, "unit": result.get("unit"), "cores": cores, "kind": None,
                                       "sample": "not timed: cpu_baseline runs on rank 0 at N=1 only" if N > 1 else "not timed (--no-cpu)"}
    late = {}
    o = others.get("beltCTR")
    if o:
        flat["beltCTR_GiBps"] = o["value"]
        flat["beltCTR_frac"] = o["roofline"]["frac"]
        flat["beltCTR_lds_frac"] = o["value"] / N / LDS_CTR_CEIL_GIBPS
        late["beltCTR_ms"] = o["roofline"]["avg_launch_ms"]
        if "cpu_baseline" in o:
            cb["beltCTR_GiBps"] = o["cpu_baseline"]["value"]
            cb["beltCTR_GiBps_single_thread"] = o["cpu_baseline"].get("single_thread")
            cb["beltCTR_scaling_over_single_thread"] = o["cpu_baseline"].get("scaling_over_single_thread")
    o = others.get("bignVerify")
    if o:
        flat["bignVerify_sigs_per_s"] = o["value"]
        flat["bignVerify_frac"] = o["roofline"]["frac"]
        late["bignVerify_ms"] = o["roofline"]["avg_batch_ms"]
        late["bignVerify_clock_ghz"] = o["roofline"].get("shader_clock_ghz_under_kernels")
        late["bignVerify_frac_at_clock"] = o["roofline"].get("frac_at_measured_clock")
        late["bignVerify_verdicts_ok"] = o["verdicts_as_expected"]
        if "cpu_baseline" in o:
            cb["bignVerify_sigs_per_s"] = o["cpu_baseline"]["value"]
            cb["bignVerify_sigs_per_s_single_thread"] = o["cpu_baseline"].get("single_thread")
            cb["bignVerify_scaling_over_single_thread"] = o["cpu_baseline"].get("scaling_over_single_thread")
    o = others.get("bash512_beltMAC")
    if o:
        flat["mixed_msgs_per_s"] = o["value"]
        flat["mixed_frac"] = o["roofline"]["frac"]
        late["mixed_frac_sum_of_parts"] = o["roofline"]["frac_sum_of_parts"]
        late["mixed_overlap_got"] = o["roofline"]["overlap_got"]
        if "cpu_baseline" in o:
            cb["mixed_msgs_per_s"] = o["cpu_baseline"]["value"]
            cb["mixed_msgs_per_s_single_thread"] = o["cpu_baseline"].get("single_thread")
            cb["mixed_scaling_over_single_thread"] = o["cpu_baseline"].get("scaling_over_single_thread")
    o = others.get("bignSign2")
    if o:
        late["bignSign2_sigs_per_s"] = o["value"]
        late["bignSign2_frac"] = o["roofline"]["frac"]
    # the strong (fixed-N) reading, SURVEY 8e: predictions from one GPU at N = 1, measured speedups at N > 1 -- the 8-way keys first
    strong_order = ([f"strong_pred_8_{w}" for w in ("bashF", "ctr", "verify", "mixed")] if N == 1
                    else [f"strong_speedup_{w}" for w in ("bashF", "ctr", "verify", "mixed")])
    for k in strong_order:                # (always present, None when the workload was not run: the key order is part of the contract)
        flat[k] = strong.get(k)
    if cb.get("kind") is not None:       # scalars the record must keep first (the driver keeps ~24 entries, strings cut at 120 characters)
        first = ("value", "unit", "cores", "kind", "sample", "single_thread", "scaling_over_single_thread", "spin_scaling", "cpu_count",
                 "affinity", "cgroup_quota_cpus")
        mid = [k for k in cb if k.startswith(("beltCTR_", "bignVerify_", "mixed_"))]
        ordered = {k: cb.get(k) for k in first}
        ordered.update({k: cb[k] for k in mid})
        ordered.update({k: v for k, v in cb.items() if k not in ordered})
        cb = result["cpu_baseline"] = ordered
    flat["n_ranks_seen"] = int(dist.sum(1.0))
    flat["n_devices_distinct"] = n_distinct
    # N > 1: the line explains itself -- rank 0 alone beforehand, each rank's own rate, clocks
    for k in ("weak_efficiency", "solo_value", "clock_ghz_min", "clock_ghz_max", "per_rank_value_min", "per_rank_value_max"):
        flat[k] = diag.get(k)
    # measured VALU utilisation (SQ counters, replayed from the committed profile of this round like `traffic`)
    for key, kern in (("valu_busy_bashF", "bashF_tile_kernel"), ("valu_busy_beltCTR", "beltCTR_blocks_kernel"),
                      ("valu_busy_bign_main", "bign_main_kernel<8"), ("valu_busy_fused", "hash_mac_fused_kernel<8, true, true"),
                      ("valu_busy_bign_mulbase", "bign_mulbase_lds_kernel<8"), ("valu_busy_bign_onekey", "bign_onekey_kernel<8"),
                      ("valu_busy_belt_hash_long", "belt_hash_long"), ("valu_busy_bash_ragged", "bash_ragged_kernel")):
        late[key] = pmc_valu(kern)
    flat["avg_launch_ms"] = rf0.get("avg_launch_ms", rf0.get("avg_batch_ms"))
    flat["kernel"] = rf0.get("kernel", rf0.get("kernels"))
    flat.update(late)
    flat["n_devices_visible"] = dist.ndev
    for k in ("per_rank_kernel_rate_min", "per_rank_kernel_rate_max"):
        flat[k] = diag.get(k)
    for k, v in strong.items():          # the rest of the strong-split record (2- / 4-way predictions, the times behind them, N > 1 rates)
        flat.setdefault(k, v)
    rest = {}
    for k, v in rf0.items():              # what is left of the headline kernel's own object: scalars stay, prose / nested move out
        if k in flat:
            continue
        if isinstance(v, (int, float, bool)) or v is None:
            flat[k] = v
        else:
            rest[k] = v
    if rest:
        others.setdefault("headline_detail", {}).update(rest)
    result["roofline"] = flat
    result["others"] = others
    result["host"] = {"cpu_count": hc["cpu_count"], "cpus": hc, "device": torch.cuda.get_device_name(torch.cuda.current_device()),
                      "engine": eng.version()}
    if dist.rank == 0:
        print(json.dumps(result))
    dist.close()


if __name__ == "__main__":
    main()
