"""bench leg: bash512 + beltMAC over 4 KiB messages (BASELINE configs[4])"""
import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, args, K, N, H, kw = c.dist, c.eng, c.args, c.K, c.N, c.H, c.kw
    c0, others, rates, strong, hc, do_cpu, strong_leg = c.c0, c.others, c.rates, c.strong, c.hc, c.do_cpu, c.strong_leg
    n, ml = 1 << 21, 4096                                      # the weak leg: 2^24 / 8 messages per GPU
    # the FIXED job of configs[4]: 2^24 x 4 KiB = 64 GiB, resident on this card when it fits (288 GB: four times over); the weak
    # leg runs over its first 2^21 messages, the whole job is `bash512_beltMAC_2p24` and the total of the strong split
    n_all = n if args.headline_only else STRONG_TOTALS["mixed"]      # (--headline-only: the 2^21-message launches and nothing else)
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
    pmc = pmc_headline("mixed", n)
    others["bash512_beltMAC"]["roofline"]["valu_busy"] = pmc.get("valu_busy") if pmc else None
    others["bash512_beltMAC"]["roofline"]["traffic"] = pmc.get("hbm_bytes_per_launch") if pmc else None
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
        from .cpu import cpu_baseline
        others["bash512_beltMAC"]["cpu_baseline"] = cpu_baseline("mixed", hc)
    del msgs, dig, tag, dig_all, tag_all

