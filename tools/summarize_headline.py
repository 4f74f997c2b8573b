#!/usr/bin/env python3
"""Condense gpurun_out/prof_headline (tools/profile_headline.sh) into <tag>_pmc_headline.json: per BASELINE leg the counters of the
ONE launch shape bench.py times -- rows are grouped by (kernel, grid size), never averaged across launch sizes -- plus the
per-kernel stats CSV of each leg.  bench_legs/common.py pmc_headline() replays `launches[leg]` when (and only when) `units` is the
number of units the run's own timed launch processed.    usage: python tools/summarize_headline.py r06"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_headline")
DST = os.environ.get("PROF_DST", os.path.join(ROOT, "profiles"))
N_CU, N_SE, N_SIMD = 256, 32, 1024

# leg -> (prefix of the kernel that carries it, units per headline launch, algorithmic HBM bytes per launch)
LEGS = {
    "bashF": ("bashF_tile_kernel", 1 << 20, 384 * (1 << 20)),
    "ctr": ("beltCTR_blocks_kernel", 1 << 30, 32 * (1 << 30)),
    "verify": ("bign_main29_kernel", 1 << 18, 148 * (1 << 18)),      # (round 6: the 29-bit one-lane kernel carries 2^18)
    "mixed": ("hash_mac_fused_kernel", 1 << 21, (4096 + 72) * (1 << 21)),
}


def short(name):
    return name.split("(")[0].replace("void ", "").replace("bee2hip::", "").strip()


def groups(path):
    """{(kernel, grid): {counter: mean, duration_ns: mean, launches: n, workgroup, ...}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        if any(x in k[0] for x in ("rocclr", "at::", "elementwise", "b2h_clock")):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = {"workgroup": int(r["Workgroup_Size"]), "lds_block_size_static": int(r["LDS_Block_Size"]),
                   "vgpr_count_rocprof": int(r["VGPR_Count"]), "scratch": int(r["Scratch_Size"])}
    out = {}
    for k, v in acc.items():
        ncounters = max(1, len([c for c in v if c != "duration_ns"]))
        out[k] = dict(meta[k], launches=len(v["duration_ns"]) // ncounters, **{c: sum(x) / len(x) for c, x in v.items()})
    return out


def pick(gr, prefix):
    """the (kernel, grid) group with this name prefix that took the most time in the run = the headline launch shape"""
    cand = [(v["duration_ns"] * v["launches"], k) for k, v in gr.items() if k[0].startswith(prefix)]
    return max(cand)[1] if cand else None


def main(tag):
    os.makedirs(DST, exist_ok=True)
    try:
        commit = os.environ.get("PROF_COMMIT") or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "?"
    out = {"commit": commit, "command": "rocprofv3 <pass> -- python bench.py --only <leg> --headline-only --no-cpu --steps 10 --warmup 2",
           "passes": {"stats": "--kernel-trace --stats", "fetch": "--pmc FETCH_SIZE", "write": "--pmc WRITE_SIZE",
                      "sq1": "--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS",
                      "sq2": "--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"},
           "definitions": {
               "hbm_bytes_per_launch": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: FETCH_SIZE counts half the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md)",
               "valu_busy": "SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 SEs x 1024 SIMDs): VALU instructions executing per SIMD (1.0 = one pipe never idle; up to 2.0 where half- and full-rate instructions of different wavefronts overlap)",
               "lds_array_busy": "SQ_LDS_IDX_ACTIVE / (256 CUs x SQ_BUSY_CYCLES / 32): share of cycles a CU's LDS array is serving an indexed access",
               "lds_inst_busy": "SQ_ACTIVE_INST_LDS x 4 / (SQ_BUSY_CYCLES / 32 x 1024): LDS instructions executing per SIMD (a different normalisation: per SIMD, not per CU array)"},
           "launches": {}, "all_groups": {}}
    for leg, (prefix, units, alg) in LEGS.items():
        files = {p: os.path.join(SRC, f"{leg}_{p}", "b_counter_collection.csv") for p in ("fetch", "write", "sq1", "sq2")}
        gr = {p: groups(f) for p, f in files.items() if os.path.exists(f)}
        if not gr:
            continue
        st = glob.glob(os.path.join(SRC, f"{leg}_stats", "b_kernel_stats.csv"))
        stats = {}
        if st:
            shutil.copy(st[0], os.path.join(DST, f"{tag}_kernel_stats_{leg}.csv"))
            for r in csv.DictReader(open(st[0])):
                stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
        e = {"units": units, "algorithmic_bytes_per_launch": alg}
        key = None
        for p in ("sq1", "sq2", "fetch", "write"):
            if p in gr:
                key = pick(gr[p], prefix)
                if key:
                    break
        if not key:
            continue
        e.update(kernel=key[0], grid=key[1])
        g = lambda p: gr.get(p, {}).get(key)  # noqa: E731
        fe, wr, s1, s2 = g("fetch"), g("write"), g("sq1"), g("sq2")
        if fe and wr:
            hbm = (2 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024
            e.update(FETCH_SIZE_KiB=fe["FETCH_SIZE"], WRITE_SIZE_KiB=wr["WRITE_SIZE"], hbm_bytes_per_launch=hbm, traffic_over_algorithmic=hbm / alg,
                     launches_profiled=fe["launches"], workgroup=fe["workgroup"])
        if s1 and s1.get("SQ_BUSY_CYCLES"):
            cyc = s1["SQ_BUSY_CYCLES"] / N_SE
            e.update(valu_busy=s1["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD), valu_insts_per_wave=s1["SQ_INSTS_VALU"] / max(1.0, s1["SQ_WAVES"]),
                     lds_insts_per_wave=s1["SQ_INSTS_LDS"] / max(1.0, s1["SQ_WAVES"]), waves=s1["SQ_WAVES"], busy_cycles_per_se=cyc,
                     wave_cycles_per_simd_cycle=s1["SQ_WAVE_CYCLES"] * 4.0 / (cyc * N_SIMD), duration_ns_under_sq1=s1["duration_ns"])
        if s2 and s2.get("SQ_BUSY_CYCLES"):
            cyc = s2["SQ_BUSY_CYCLES"] / N_SE
            e.update(lds_array_busy=s2["SQ_LDS_IDX_ACTIVE"] / (N_CU * cyc), lds_inst_busy=s2["SQ_ACTIVE_INST_LDS"] * 4.0 / (cyc * N_SIMD),
                     lds_bank_conflict_cycles=s2["SQ_LDS_BANK_CONFLICT"], salu_insts=s2.get("SQ_INSTS_SALU"),
                     wait_inst_any_per_active=s2["SQ_WAIT_INST_ANY"] / max(1.0, s2["SQ_ACTIVE_INST_ANY"]))
        sk = next((v for k, v in stats.items() if k.startswith(prefix)), None)
        if sk:
            e.update(avg_duration_ns_kernel_trace=sk["avg_ns"], calls_kernel_trace=sk["calls"], pct_of_leg_kernel_time=sk["pct"])
        out["launches"][leg] = e
        # every (kernel, grid) group of the leg's SQ passes: the other kernels of the leg (verification is five) and proof of one launch size
        allg = {}
        for k, v in sorted(gr.get("sq1", {}).items(), key=lambda kv: -kv[1]["duration_ns"] * kv[1]["launches"]):
            cyc = v["SQ_BUSY_CYCLES"] / N_SE if v.get("SQ_BUSY_CYCLES") else None
            w = gr.get("sq2", {}).get(k)
            allg[f"{k[0]} @grid {k[1]}"] = {
                "launches": v["launches"], "duration_ns": v["duration_ns"],
                "valu_busy": v["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD) if cyc else None,
                "valu_insts_per_wave": v["SQ_INSTS_VALU"] / max(1.0, v["SQ_WAVES"]),
                "lds_array_busy": (w["SQ_LDS_IDX_ACTIVE"] / (N_CU * w["SQ_BUSY_CYCLES"] / N_SE)) if w and w.get("SQ_BUSY_CYCLES") else None}
        out["all_groups"][leg] = allg
    with open(os.path.join(DST, f"{tag}_pmc_headline.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["launches"], indent=1))
    for leg, a in out["all_groups"].items():
        print(leg, list(a)[:12])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
