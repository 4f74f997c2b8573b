"""The bench's output: ONE strict-JSON headline line of at most LINE_MAX characters (the driver's record keeps the last ~8 KB of
stdout and parses its last line), one human-readable figure per line before it (the shape of bee2's own benches,
test/crypto/bash_bench.c:63-74), and everything else -- every leg's own object, sweeps, host-pointer rates, the strong split's
times, prose -- in a side file (gpurun_out/bench_detail.json).  Pure functions of dictionaries: tests/test_bench_launch.py builds
lines here on CPU and checks length, strictness and key order."""
import json
import math
import os

LINE_MAX = 8000
WORKLOAD_KEYS = ("bashF", "ctr", "verify", "mixed")
# `roofline`: 24 flat scalars, in this order, for every N.  Slots 15-18 are the fixed-N (strong) reading of SURVEY 8e:
# strong_pred_8_* at N = 1 (predicted from one GPU), strong_speedup_* at N > 1 (measured).
ROOFLINE_HEAD = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                 "beltCTR_GiBps", "beltCTR_frac", "bignVerify_sigs_per_s", "bignVerify_frac", "mixed_msgs_per_s", "mixed_frac")
ROOFLINE_TAIL = ("n_ranks_seen", "n_devices_distinct", "valu_busy", "frac_2p22", "beltCTR_lds_frac", "weak_efficiency")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "single_thread", "scaling_over_single_thread", "spin_scaling", "cpu_count",
            "beltCTR_GiBps", "bignVerify_sigs_per_s", "mixed_msgs_per_s", "impl")
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def strong_keys(n_ranks):
    return tuple((f"strong_pred_8_{w}" if n_ranks == 1 else f"strong_speedup_{w}") for w in WORKLOAD_KEYS)


def roofline_keys(n_ranks):
    return ROOFLINE_HEAD + strong_keys(n_ranks) + ROOFLINE_TAIL


def clean(x, digits=8, maxstr=None):
    """JSON-strict copy: NaN / inf -> None, floats to `digits` significant digits, numpy scalars to Python, long strings cut"""
    if isinstance(x, dict):
        return {str(k): clean(v, digits, maxstr) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [clean(v, digits, maxstr) for v in x]
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, str):
        return x if maxstr is None or len(x) <= maxstr else x[: maxstr - 3] + "..."
    if hasattr(x, "item"):                                  # numpy / torch scalar
        return clean(x.item(), digits, maxstr)
    return clean(str(x), digits, maxstr)


def build(result, others, strong, diag, n_ranks, n_ranks_seen, n_distinct, lds_ctr_ceil_gibps, detail_path=None):
    """(line, detail): the headline object and the side file's object.  `result` is the headline leg's own record (top-level
    fields + its `roofline` + `cpu_baseline`), `others` every other leg's record."""
    rf0 = result.get("roofline") or {}
    flat = {k: None for k in roofline_keys(n_ranks)}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        flat[k] = rf0.get(k)
    flat["kernel"] = rf0.get("kernel", rf0.get("kernels"))
    flat["avg_launch_ms"] = rf0.get("avg_launch_ms", rf0.get("avg_batch_ms"))
    flat["valu_busy"] = rf0.get("valu_busy")
    flat["frac_2p22"] = rf0.get("frac_2p22")
    cb = dict(result.get("cpu_baseline") or {})
    o = others.get("beltCTR")
    if o:
        flat["beltCTR_GiBps"] = o["value"]
        flat["beltCTR_frac"] = o["roofline"]["frac"]
        flat["beltCTR_lds_frac"] = o["value"] / n_ranks / lds_ctr_ceil_gibps
        if "cpu_baseline" in o:
            cb["beltCTR_GiBps"] = o["cpu_baseline"]["value"]
    o = others.get("bignVerify")
    if o:
        flat["bignVerify_sigs_per_s"] = o["value"]
        flat["bignVerify_frac"] = o["roofline"]["frac"]
        if "cpu_baseline" in o:
            cb["bignVerify_sigs_per_s"] = o["cpu_baseline"]["value"]
    o = others.get("bash512_beltMAC")
    if o:
        flat["mixed_msgs_per_s"] = o["value"]
        flat["mixed_frac"] = o["roofline"]["frac"]
        if "cpu_baseline" in o:
            cb["mixed_msgs_per_s"] = o["cpu_baseline"]["value"]
    for k in strong_keys(n_ranks):                          # always present, None when the workload was not run
        flat[k] = strong.get(k)
    flat["n_ranks_seen"] = n_ranks_seen
    flat["n_devices_distinct"] = n_distinct
    flat["weak_efficiency"] = diag.get("weak_efficiency")
    line = {k: result.get(k) for k in TOP_KEYS}
    line["roofline"] = flat
    if cb.get("kind") is None:
        cb = {"value": None, "unit": result.get("unit"), "cores": cb.get("cores"), "kind": None,
              "sample": "not timed: cpu_baseline runs on rank 0 at N=1 only" if n_ranks > 1 else "not timed (--no-cpu)"}
    line["cpu_baseline"] = {k: cb.get(k) for k in CPU_KEYS if k in cb or k in ("value", "unit", "cores", "kind", "sample")}
    if n_ranks > 1:                                         # the N > 1 line explains itself: rank 0 alone beforehand, each rank's own rate, clocks
        line["ranks"] = {k: diag.get(k) for k in ("solo_value", "weak_efficiency", "per_rank_value_min", "per_rank_value_max",
                                                  "clock_ghz_min", "clock_ghz_max")}
    if detail_path:
        line["detail"] = detail_path
    line = clean(line, maxstr=300)
    line["roofline"] = clean(line["roofline"], maxstr=60)
    line["cpu_baseline"] = clean(line["cpu_baseline"], maxstr=120)
    detail = clean({"headline": result, "others": others, "strong": strong, "ranks": diag}, digits=9)
    return line, detail


def dumps(line):
    """the line as printed: strict JSON (no NaN / Infinity), one line, at most LINE_MAX characters -- asserted, not hoped for"""
    s = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    if len(s) > LINE_MAX or "\n" in s:
        raise AssertionError(f"bench line is {len(s)} characters (limit {LINE_MAX}): move fields to the detail file")
    return s


def fmt(v, unit):
    if v is None:
        return "n/a"
    for scale, pre in ((1e9, "G"), (1e6, "M"), (1e3, "k")):
        if abs(v) >= scale and unit.split("/")[0] not in ("GiB", "GB"):
            return f"{v / scale:.3f} {pre} {unit}"
    return f"{v:.3f} {unit}"


def figures(line, detail):
    """one figure per line, for a reader of the log (the JSON line carries the same numbers)"""
    rf, cb = line["roofline"], line["cpu_baseline"]
    n = line.get("n_gpus")
    out = [f"[bench] {line['metric']} on {n} GPU(s): {fmt(line['value'], line['unit'])}; kernel {rf.get('kernel')}: "
           f"{rf.get('avg_launch_ms')} ms per launch = {rf.get('achieved')} {rf.get('unit')} of {rf.get('peak')} ({rf.get('frac')})"]
    for tag, key, unit, frac in (("beltCTR", "beltCTR_GiBps", "GiB/s", "beltCTR_frac"), ("bignVerify256", "bignVerify_sigs_per_s", "verifies/s", "bignVerify_frac"),
                                 ("bash512+beltMAC", "mixed_msgs_per_s", "messages/s", "mixed_frac")):
        if rf.get(key) is not None:
            cpu = cb.get(key)
            out.append(f"[bench] {tag}: {fmt(rf[key], unit)} (roofline fraction {rf.get(frac)})" + (f"; host cores ({cb.get('cores')}): {fmt(cpu, unit)}" if cpu else ""))
    if cb.get("value") is not None:
        out.append(f"[bench] host {cb.get('impl')}: {fmt(cb['value'], cb.get('unit') or '')} on {cb.get('cores')} threads, {fmt(cb.get('single_thread'), cb.get('unit') or '')} on one")
    sk = [k for k in rf if k.startswith("strong_") and rf[k] is not None]
    if sk:
        out.append("[bench] fixed-N split, 8 ways: " + ", ".join(f"{k.split('_')[-1]} x{rf[k]:.2f}" for k in sk)
                   + (" (predicted from this GPU)" if sk[0].startswith("strong_pred") else " (measured)"))
    return out


def write_detail(detail, root):
    """gpurun_out/bench_detail.json under the repo (merged back from the GPU box by gpurun); returns the relative path or None"""
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, "bench_detail.json")
        with open(p, "w") as fh:
            json.dump(detail, fh, indent=1, allow_nan=False)
        return "gpurun_out/bench_detail.json"
    except OSError:
        return None
