"""bench leg: beltCTR over a 16 GiB stream (BASELINE configs[2])"""
import ctypes

import numpy as np
import torch

from bee2_amd import shard
from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, args, K, N, H, kw = c.dist, c.eng, c.args, c.K, c.N, c.H, c.kw
    c0, others, rates, strong, hc, do_cpu, strong_leg = c.c0, c.others, c.rates, c.strong, c.hc, c.do_cpu, c.strong_leg
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
    pmc = pmc_headline("ctr", nb)                            # replayed only if the profiled launch had this many blocks
    ctr_traffic, ctr_traffic_src = (pmc["hbm_bytes_per_launch"], pmc["source"]) if pmc else (None, None)
    others["beltCTR"] = {
        "metric": "beltCTR GiB/s", "value": N * nbytes * kc / el / 2 ** 30, "unit": "GiB/s", "steps": kc,
        "ms_per_step": el / kc * 1e3,
        "config": {"workload": f"beltCTR bulk encrypt, {nbytes / 2**30:.1f} GiB stream per GPU, one key (BASELINE configs[2])"},
        "roofline": {"kernel": "beltCTR_blocks_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": ctr_traffic, "traffic_source": ctr_traffic_src,
                     "avg_launch_ms": ms_launch, "valu_busy": pmc.get("valu_busy") if pmc else None,
                     "lds_array_busy": pmc.get("lds_array_busy") if pmc else None,
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
        from .cpu import cpu_baseline
        others["beltCTR"]["cpu_baseline"] = cpu_baseline("ctr", hc)
    del buf

