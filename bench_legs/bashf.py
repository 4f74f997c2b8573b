"""bench leg: bashF over 2^20 states (BASELINE configs[1], the headline)"""
import ctypes

import torch

from bee2_amd import shard  # noqa: F401
from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, args, K, W, N, result = c.dist, c.eng, c.args, c.K, c.W, c.N, c.result
    others, rates, strong, diag, hc, do_cpu, strong_leg = c.others, c.rates, c.strong, c.diag, c.hc, c.do_cpu, c.strong_leg
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
    pmc = pmc_headline("bashF", n)                           # a replay, and only of this very launch (same kernel, 2^20 states)
    traffic, traffic_src = (pmc["hbm_bytes_per_launch"], pmc["source"]) if pmc else (None, None)
    # the shader clock the chip sustained (power: ~1.8-1.9 GHz under this kernel is normal; one box of the pool ran everything
    # at half speed): measured on every rank, outside the timed region
    try:
        ghz_under, ghz_idle = shader_clock_under(step, ms_launch)
        clock_note = None if ghz_under else "libb2hprobe.so not built"
    except Exception as e:                                    # the probe is a diagnostic, never a reason to fail the bench
        ghz_under, ghz_idle, clock_note = None, None, repr(e)
    clocks = [g for g in dist.gather(ghz_under) if g]
    result.update({
        "metric": "bashF perms/s", "value": value, "unit": "perms/s", "n_gpus": N, "steps": K, "warmup": W,
        "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "bashF batch: 2^20 independent 192-byte sponge states per GPU (BASELINE configs[1])",
                   "states_per_gpu": n, "parallelism": f"dp{N}: index-sharded, no data-path collective; value = WEAK reading (every rank the full batch); "
                                  "fixed-N split = roofline.strong_pred_8_* (N=1, predicted) / strong_speedup_* (N>1, measured)"},
        "roofline": {"kernel": "bashF_tile_kernel<0, 2, 124, 6, 3>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_launch_ms": ms_launch, "algorithmic_bytes_per_launch": BASHF_BYTES * n,
                     "shader_clock_ghz_under_kernel": ghz_under, "shader_clock_ghz_idle": ghz_idle,
                     "valu_busy": pmc.get("valu_busy") if pmc else None},
    })
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
        # A 2^17-state share is a 17 us kernel: an eager Python loop measures the host's launch rate there (~21 us per step).  The strong
        # legs therefore time REPLAYS OF CAPTURED GRAPHS of K launches -- at N = 1 (the prediction) and at N > 1 (the measurement) alike;
        # the eager figures stay in the detail file as strong_*_bashF_eager.
        strong_leg("bashF", n, bash_unit, K, graph=True)
        if N == 1:
            for g, m in strong_shares(n).items():
                t_eager = event_ms(bash_unit(0, m), K)
                strong[f"strong_ms_share{g}_bashF_eager"] = t_eager
                strong[f"strong_pred_{g}_bashF_eager"] = ms_launch / t_eager
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
        from .cpu import cpu_baseline
        result["cpu_baseline"] = cpu_baseline("bashF", hc)
    del st

