"""HBM traffic of the headline launch measured IN THIS RUN: the bench re-runs its headline leg alone under `rocprofv3 --pmc` -- one
pass for FETCH_SIZE, one for WRITE_SIZE (they do not fit one pass), no trace domain beside the counters -- and reads the per-launch
averages of the very kernel and grid it timed.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on
gfx950.  Rank 0 at N = 1 only, after the timed legs (nothing here is inside a timed region); any failure (no rocprofv3, a timeout,
an unexpected CSV) returns None and the line falls back to the committed profile of the same launch, or null."""
import csv
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_pass(counters, leg, timeout_s):
    """one rocprofv3 --pmc pass over the headline-only run of `leg`; -> {counter: {(kernel, grid): [values per launch]}}"""
    counters = counters.split()
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tmp = tempfile.mkdtemp(prefix="b2h_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", tmp, "-o", "b", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--only", leg, "--headline-only", "--no-cpu", "--no-live-pmc",
               "--steps", "5", "--warmup", "2"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env["TMPDIR"] = "/tmp"
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return None
        path = None
        for d, _, files in os.walk(tmp):
            if "b_counter_collection.csv" in files:
                path = os.path.join(d, "b_counter_collection.csv")
        if not path:
            return None
        rows = {c: {} for c in counters}
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] not in rows:
                continue
            key = (row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bee2hip::", "").strip(), int(row["Grid_Size"]))
            rows[row["Counter_Name"]].setdefault(key, []).append(float(row["Counter_Value"]))
        return rows
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def headline_traffic(leg, kernel_prefix, grid, timeout_s=90):
    """-> {"hbm_bytes_per_launch", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "launches", "source"} for launches of `kernel_prefix` with
    `grid` work-items, or None"""
    fe = _one_pass("FETCH_SIZE", leg, timeout_s)
    if not fe:
        return None
    wr = _one_pass("WRITE_SIZE", leg, timeout_s)
    if not wr:
        return None
    pick = lambda rows: next((v for (k, g), v in rows.items() if k.startswith(kernel_prefix) and g == grid), None)  # noqa: E731
    f, w = pick(fe["FETCH_SIZE"]), pick(wr["WRITE_SIZE"])
    if not f or not w:
        return None
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out = {"hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "launches": min(len(f), len(w)),
           "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH x 2) over bench.py --only "
                     f"{leg} --headline-only"}
    # VALU utilisation of the same launch (north_star's "VALU integer-op utilisation"): SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32
    # shader engines x 1024 SIMDs) = VALU instructions executing per SIMD -- a third pass; its absence does not void the traffic figure
    sq = _one_pass("SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES", leg, timeout_s)
    if sq:
        a, b = pick(sq["SQ_ACTIVE_INST_VALU"]), pick(sq["SQ_BUSY_CYCLES"])
        if a and b and sum(b) > 0:
            out["valu_busy"] = (sum(a) / len(a)) * 4.0 / ((sum(b) / len(b)) / 32.0 * 1024.0)
    return out
