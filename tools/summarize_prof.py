#!/usr/bin/env python3
"""Condense gpurun_out/prof (rocprofv3 output of tools/profile_round.sh) into the small,
committed summaries under profiles/.   usage: python tools/summarize_prof.py r01"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.environ.get("PROF_DST", os.path.join(ROOT, "profiles"))     # (on the GPU box: a directory under gpurun_out/, copied into profiles/ afterwards)


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("bee2hip::", "")
    return n.strip()


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                   "lds_block_size_static": int(r["LDS_Block_Size"]), "vgpr_count_rocprof": int(r["VGPR_Count"]),
                   "sgpr_count": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"])}
    out = {}
    for k, v in d.items():
        if "rocclr" in k or "at::" in k or "elementwise" in k:
            continue
        out[k] = dict(meta[k], **{c: sum(x) / len(x) for c, x in v.items()})
    return out


def main(tag):
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "stats", "bench_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats.csv"))
    hb = os.path.join(SRC, "stats_bashF", "bench_kernel_stats.csv")
    if os.path.exists(hb):
        shutil.copy(hb, os.path.join(DST, f"{tag}_kernel_stats_bashF_only.csv"))
    summary = {"command": "rocprofv3 --kernel-trace --stats / --pmc <set> -- python bench.py --steps 20 --warmup 3 --no-cpu "
                          "(PMC passes: --ctr-gib 4; FETCH/WRITE passes: --only bashF,ctr)",
               "valu_busy": "kernels[*].valu_busy = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs), pass pmc_sq1: the average number of VALU instructions executing per SIMD (1.0 = one pipe never idle; up to 2.0 when half- and full-rate instructions of different wavefronts overlap)",
               "note": "counter values are per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts "
                       "half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> doubled in hbm_bytes_per_launch; "
                       "vgpr_count_rocprof is what rocprofv3 prints, which on gfx950 is HALF the allocated VGPRs "
                       "(bashF: ISA .amdhsa_next_free_vgpr 113 -> 120 allocated -> 60 here; bign_main<8>: 166 -> 168 -> 84); "
                       "lds_block_size_static excludes dynamic LDS (bashF uses 52 KiB of it per workgroup)"}
    for f in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq1", "pmc_sq2"):
        p = os.path.join(SRC, f, "bench_counter_collection.csv")
        if os.path.exists(p):
            summary[f] = agg(p)
    with open(os.path.join(DST, f"{tag}_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    # per-kernel HBM traffic for bench.py's roofline.traffic (pmc_traffic() replays it, labelled with file + commit)
    import subprocess
    try:
        commit = os.environ.get("PROF_COMMIT") or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "?"
    kernels = {}
    for kern in ("bashF_tile_kernel", "bashF_batch_kernel", "beltCTR_blocks_kernel", "bign_pubkey_val_kernel"):
        pickk = lambda d: next(((k, v) for k, v in d.items() if k.startswith(kern)), (None, None))  # noqa: E731
        kf, fe = pickk(summary.get("pmc_FETCH_SIZE", {}))
        kw, wr = pickk(summary.get("pmc_WRITE_SIZE", {}))
        if fe and wr:
            kernels[kf] = {"FETCH_SIZE_KiB": fe["FETCH_SIZE"], "WRITE_SIZE_KiB": wr["WRITE_SIZE"],
                           "hbm_bytes_per_launch": (2 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024,
                           "avg_duration_ns_under_pmc": fe["duration_ns"],
                           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x 2 (gfx950)"}
    # VALU utilisation per kernel from the SQ pass (north_star: "VALU integer-op utilisation"): SQ_ACTIVE_INST_VALU counts, per SIMD, the
    # quad-cycles a VALU instruction is executing; SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): cycles per SE x 1024
    # SIMDs are the SIMD-cycles there were.  The counter sums over wavefronts, so it is the average number of VALU instructions
    # EXECUTING per SIMD: 1.0 = one pipe never idle (bign_main_kernel: multiply-adds and carries do not overlap), up to 2.0 where
    # half-rate and full-rate instructions of different wavefronts run side by side (bashF_tile_kernel: 1.75).
    sq1, sq2 = summary.get("pmc_sq1", {}), summary.get("pmc_sq2", {})
    for k, v in sq1.items():
        busy, act = v.get("SQ_BUSY_CYCLES"), v.get("SQ_ACTIVE_INST_VALU")
        if not busy or act is None:
            continue
        e = kernels.setdefault(k, {})
        e["valu_busy"] = act * 4.0 / (busy / 32.0 * 1024.0)
        e["valu_insts_per_wave"] = v.get("SQ_INSTS_VALU", 0.0) / max(1.0, v.get("SQ_WAVES", 1.0))
        e["waves"] = v.get("SQ_WAVES")
        e["wave_cycles_per_simd_cycle"] = v.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / (busy * 32.0)
        e["lds_insts_per_wave"] = v.get("SQ_INSTS_LDS", 0.0) / max(1.0, v.get("SQ_WAVES", 1.0))
        e["duration_ns_under_sq_pass"] = v.get("duration_ns")
        w = sq2.get(k)
        if w and w.get("SQ_BUSY_CYCLES"):
            e["lds_busy"] = w.get("SQ_ACTIVE_INST_LDS", 0.0) * 4.0 / (w["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
            e["lds_bank_conflict_cycles"] = w.get("SQ_LDS_BANK_CONFLICT")
        e.setdefault("note_valu", "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024); rocprofv3 --pmc, own pass")
    summary["kernels"] = kernels
    summary["commit"] = commit
    with open(os.path.join(DST, f"{tag}_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    for kern, fname, alg in (("bashF_tile_kernel" if tag != "r01" else "bashF_batch_kernel", f"{tag}_bashF_pmc.json", 384 * (1 << 20)),):
        pick = lambda d: next((v for k, v in d.items() if k.startswith(kern)), None)  # noqa: E731
        fe = pick(summary.get("pmc_FETCH_SIZE", {}))
        wr = pick(summary.get("pmc_WRITE_SIZE", {}))
        if fe and wr:
            hbm = (2 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024
            with open(os.path.join(DST, fname), "w") as fh:
                json.dump({"kernel": kern, "FETCH_SIZE_KiB": fe["FETCH_SIZE"], "WRITE_SIZE_KiB": wr["WRITE_SIZE"],
                           "fetch_correction": "x2 (gfx950 wide coalesced reads)", "hbm_bytes_per_launch": hbm,
                           "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
                           "avg_duration_ns_under_pmc": fe["duration_ns"]}, fh, indent=1)
    print(open(os.path.join(DST, f"{tag}_kernel_stats.csv")).read()[:3000])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
