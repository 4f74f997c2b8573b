"""What every bench leg shares: the roofline constants, the process group (one process per GPU, RCCL), the timed
region (barrier + synchronize on both sides, hipEvents on the launch stream inside it), the fixed-N (strong) split of SURVEY 8e,
the shader-clock probe and the replay of committed PMC counters.  bench.py drives the legs; bench_legs/line.py builds the JSON line."""
import ctypes
import json
import os
import sys
import time

import numpy as np  # noqa: F401
import torch

from bee2_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

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
        # (capturing takes the host tens of milliseconds during which the chip idles and drops its clock: the replay is warmed like
        #  the eager loop above before it is timed -- one cold replay of 20 launches read 116 us per 2^20-state launch instead of 91)
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
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


def graph_of(fn, steps):
    """`steps` calls of fn() captured into one hipGraph (on a side stream; the engine launches on torch's current stream)"""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(steps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    return g


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


def pmc_headline(leg, units):
    """Counters of the committed headline PMC passes (tools/profile_headline.sh: one `rocprofv3 --pmc` pass per counter set per
    BASELINE launch, `bench.py --only <leg> --headline-only`; FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected inside an un-profiled run, so the line REPLAYS them --
    but only when the profiled launch is the launch this run timed (same leg, same number of units per launch); anything else
    returns None and the line says null."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_headline.json")))
    if not found:
        return None
    try:
        d = json.load(open(found[-1]))
        e = d.get("launches", {}).get(leg)
        if not e or int(e.get("units", -1)) != int(units):
            return None
        return dict(e, source=f"profiles/{os.path.basename(found[-1])} @ {d.get('commit', '?')}")
    except Exception:
        return None


# ------------------------------------------------------------------------- inputs

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


class Ctx:
    """one bench run: the engine, the process group, K / W / N, the broadcast parameters and the dictionaries the legs fill"""

    def __init__(self, args, dist, eng):
        self.args, self.dist, self.eng = args, dist, eng
        self.K, self.W, self.N = args.steps, args.warmup, dist.world
        self.result = {}                       # the headline leg's line fields
        self.others = {}                       # every other leg's object (goes to the detail file)
        self.rates = {}                        # per-GPU kernel rates the mixed roofline is built from
        self.strong = {}                       # the fixed-N (strong) reading of SURVEY 8e, flat scalars
        self.diag = {}                         # N > 1 self-explanation, flat scalars
        self.H = self.kw = self.c0 = None
        self.hc, self.cores, self.do_cpu = None, 1, False

    def strong_leg(self, name, total, unit_fn, steps, to_value=1.0, t_total_ms=None, graph=False):
        """The FIXED job of `total` items (BASELINE's size) split by shard.shard_range.  unit_fn(lo, hi) returns the step over
        items [lo, hi) of the resident job.  N = 1: time rank 0's share of a 2- / 4- / 8-way split on this GPU, like the headline
        (hipEvents around `steps` launches) -> strong_pred_G_<name> = t(total) / t(total / G).  N > 1: every rank its own
        share between barriers -> strong_value_<name> (whole job, in the metric's unit) and strong_speedup_<name> = rank 0
        alone on the total / the N ranks on their shares."""
        dist, N, strong = self.dist, self.N, self.strong
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
            f_tot, f_sh, k = unit_fn(0, total), unit_fn(lo, hi), steps
            if graph:
                # launches of a few microseconds (a 2^17-state bashF share: 17 us): `steps` of them captured into ONE hipGraph per leg,
                # the replay timed between the barriers -- an eager Python loop would time the host's launch rate, not the N GPUs
                g_tot, g_sh = graph_of(f_tot, steps), graph_of(f_sh, steps)
                f_tot, f_sh, k = g_tot.replay, g_sh.replay, 1
            solo = solo_timed(dist, k, 2, f_tot)                          # rank 0 alone on the WHOLE job
            el_s = timed(dist, k, 2, f_sh)                                # every rank its share, max over ranks
            strong[f"strong_value_{name}"] = total * steps / el_s * to_value
            strong[f"strong_solo_value_{name}"] = total * steps / solo * to_value
            strong[f"strong_speedup_{name}"] = solo / el_s
