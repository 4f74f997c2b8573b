#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 5 from a bench.py JSON line.
usage: python tools/bench_table.py [profiles/r01_bench.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path):
    line = [l for l in open(path) if l.lstrip().startswith("{")][0]
    d = json.loads(line)
    o = d["others"]
    cb = d.get("cpu_baseline", {})
    g = lambda e, *ks: (lambda v: v)(__import__("functools").reduce(lambda a, k: (a or {}).get(k), ks, e))
    rows = [("metric", "GPU", "host CPU, the reference itself (`oracle/_ref`)")]
    rows.append(("bashF, 2^20 states",
                 f"**{d['value']/1e9:.2f} G perm/s** wall ({d['roofline']['achieved']/384:.2f} G/s from kernel time; "
                 f"{d['roofline']['achieved']/1e3:.2f} TB/s = **{d['roofline']['frac']:.3f}** of HBM peak; "
                 f"{d['roofline']['valu']['frac_of_simd_cycles']:.2f} of all SIMD issue cycles)",
                 f"BASH_AVX512 {cb.get('value', 0)/1e6:.0f} M / {cb.get('single_thread', 0)/1e6:.1f} M perm/s (256 threads / 1); "
                 f"BASH_64 {cb.get('bash64_all_cores', 0)/1e6:.0f} M / {cb.get('bash64_single_thread', 0)/1e6:.1f} M"))
    c = o["beltCTR"]
    rows.append(("beltCTR, 16 GiB", f"**{c['value']:.0f} GiB/s** ({c['roofline']['achieved']/1e3:.2f} TB/s = {c['roofline']['frac']:.2f} of HBM peak; "
                 f"VALU alone {c['roofline']['valu']['frac_of_simd_cycles']:.2f} of SIMD cycles)",
                 f"{g(c,'cpu_baseline','value') or 0:.2f} / {g(c,'cpu_baseline','single_thread') or 0:.3f} GiB/s"))
    v = o["bignVerify"]
    rows.append(("bign-curve256v1 verify, 2^18", f"**{v['value']/1e6:.1f} M/s** ({v['roofline']['frac']:.2f} of the measured `v_mad_u64_u32` rate)",
                 f"{(g(v,'cpu_baseline','value') or 0)/1e3:.0f} k/s (bee2's process-global curve mutex serialises threads) / "
                 f"{(g(v,'cpu_baseline','single_thread') or 0)/1e3:.1f} k/s"))
    if "bignPubkeyVal" in o:
        rows.append(("bign public-key validation, 2^24 keys", f"**{o['bignPubkeyVal']['value']/1e9:.1f} G keys/s** ({o['bignPubkeyVal']['roofline']['achieved']/1e3:.2f} TB/s = {o['bignPubkeyVal']['roofline']['frac']:.2f} of HBM peak)",
                     (f"{g(o['bignPubkeyVal'],'cpu_baseline','value')/1e6:.1f} M keys/s per thread"
                      if g(o['bignPubkeyVal'], 'cpu_baseline', 'value') else "—")))
    rows.append(("bign-curve384v1 / 512v1 verify, 2^18", f"**{o['bignVerify_l192']['value']/1e6:.1f} / {o['bignVerify_l256']['value']/1e6:.2f} M/s**",
                 f"{(g(o['bignVerify_l192'],'cpu_baseline','value') or 0)/1e3:.1f} / {(g(o['bignVerify_l256'],'cpu_baseline','value') or 0)/1e3:.1f} k/s per thread"))
    m = o["bash512_beltMAC"]
    rows.append(("bash512+beltMAC, 2^21 x 4 KiB", f"**{m['value']/1e6:.1f} M msg/s**", f"{(g(m,'cpu_baseline','value') or 0)/1e3:.0f} k msg/s"))
    b = o["belt_modes"]
    bc = b.get("cpu_baseline", {})
    rows.append(("belt ECB / CBC-decrypt / BDE, 4 GiB", f"**{b['ecb_encr']:.0f} / {b['cbc_decr']:.0f} / {b['bde_encr']:.0f} GiB/s**",
                 f"{bc.get('ecb_encr', 0):.2f} / {bc.get('cbc_decr', 0):.2f} / {bc.get('bde_encr', 0):.2f} GiB/s"))
    rows.append(("belt-sde, 512 B / 4 KiB sectors", f"**{b['sde_encr_512']:.0f} / {b['sde_encr_4096']:.0f} GiB/s** (§4.5: 2 E per block = half of ECB at best)",
                 f"{bc.get('sde_encr', 0):.2f} GiB/s"))
    w = o["belt_dwp"]
    rows.append(("belt-dwp / belt-che wrap, 4 GiB", f"**{w['value']:.0f} / {w['che_wrap']:.0f} GiB/s** (MAC alone {w['mac_only']/1024:.2f} TiB/s)",
                 f"{g(w,'cpu_baseline','value') or 0:.2f} GiB/s (64 threads)"))
    r = o["hash_ragged"]
    if "belt_hash_uniform_1000B" in r:
        rows.append(("belt-hash / bash256, 2^18 msgs x 1000 B", f"**{r['belt_hash_uniform_1000B']:.0f} / {r['bash256_uniform_1000B']:.0f} GiB/s**", "—"))
    rows.append(("ragged belt-hash / bash256, 2^16 msgs", f"**{r['belt_hash_longest_first']:.1f} / {r['bash256_longest_first']:.1f} GiB/s** (§4.7: serial-chain bound)",
                 f"{g(r,'cpu_baseline','belt_hash') or 0:.1f} / {g(r,'cpu_baseline','bash256') or 0:.1f} GiB/s (64 threads)"))
    print("| " + " | ".join(rows[0]) + " |")
    print("|---|---|---|")
    for row in rows[1:]:
        print("| " + " | ".join(row) + " |")
    print(f"\n<!-- kernel time of the headline: {d['roofline']['avg_launch_ms']*1e3:.1f} us -->")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_bench.json"))
