"""bench leg: ragged hash batches, the bsum front-end"""
import ctypes
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, K, N, others, cores, do_cpu = c.dist, c.eng, c.K, c.N, c.others, c.cores, c.do_cpu
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

