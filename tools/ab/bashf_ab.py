"""A/B of the bashF kernel variants inside ONE process on one box (run on the GPU:
python tools/ab/bashf_ab.py [variants...]).  Each variant is first checked against the oracle on a
ragged batch, then timed (hipEvents around `reps` launches) in alternation with the others."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import orclib  # noqa: E402

NAMES = {0: "r01 product: VGPR-staged load, slab store, staged order",
         1: "LDS-DMA load, slab store, staged", 2: "LDS-DMA load, direct store, staged",
         3: "no LDS (strided 16 B), staged, 4 w/SIMD", 4: "walk: DMA prefetch, slab store, staged",
         5: "walk: DMA prefetch, direct store, staged", 6: "no LDS, compact, 8 w/SIMD (64 VGPRs)",
         7: "no LDS, compact, 8 w/SIMD (59 VGPRs)",
         10: "no LDS, staged2 W=8 (97 VGPRs, 4 w/SIMD)", 11: "no LDS, staged2 W=4 (73 VGPRs, 6 w/SIMD)",
         14: "no LDS, staged2 W=2 (61 VGPRs, 8 w/SIMD)",
         16: "direct load, half-slab store, staged2 W=8", 17: "half-slab DMA load + store, staged2 W=8",
         18: "direct load, half-slab store, staged2 W=4", 19: "LDS-DMA load, slab store, staged2 W=8",
         20: "no LDS, staged, forced 5 w/SIMD (12 B spill)",
         30: "PRIO: no LDS, staged r01", 31: "PRIO: no LDS, staged2 W=8", 32: "PRIO: no LDS, staged2 W=4",
         33: "PRIO: no LDS, staged2 W=2", 34: "PRIO: direct load, half-slab store, W=4",
         35: "PRIO: LDS-DMA load, slab store, staged r01", 36: "PRIO: direct load, half-slab store, W=8",
         37: "PRIO: half-slab DMA load + store, W=4",
         50: "v36 + prio 3 until loads issued", 51: "v36 + prio 3 for load issue and store phase",
         52: "PRIO W=2 (8 w/SIMD), quarter-slab store", 53: "  + prio 3 until loads issued",
         54: "PRIO W=4 (6 w/SIMD), half-slab 6.5 KiB store", 55: "  + prio 3 until loads issued",
         56: "PRIO W=4, quarter-slab store, load prio", 57: "PRIO W=2, eighth-slab store, load prio",
         58: "PRIO W=2, direct store, load prio", 59: "PRIO r01 staged, quarter-slab store, load prio",
         60: "PRIO W=2, quarter-slab, load+store prio", 61: "PRIO W=4, half-slab, load+store prio",
         62: "PRIO W=8, quarter-slab, load+store prio", 63: "PRIO W=4, quarter-slab, load+store prio",
         70: "product (v61) + non-temporal loads", 71: "product + non-temporal stores", 72: "product + non-temporal loads and stores",
         40: "ablation: rounds only, W=8, PRIO", 41: "ablation: rounds only, W=8, no priority",
         42: "ablation: memory only (direct load, half-slab store)", 43: "ablation: memory only (direct load + store)",
         44: "ablation: rounds only, W=4, PRIO (6 w/SIMD)", 45: "ablation: rounds only, W=2, PRIO (8 w/SIMD)",
         46: "ablation: rounds only, r01 staged, PRIO"}
ABLATION = {40, 41, 42, 43, 44, 45, 46}


def main():
    variants = [int(x) for x in sys.argv[1:]] or [0, 1, 3, 4, 11, 31, 36, 53, 61, 40, 41, 42]
    eng = bee2_amd.load_experiments()
    eng.set_device(0)
    orc = orclib.load()
    tune = eng.lib.bee2hip_internal_tune
    tune.restype = ctypes.c_uint32
    n_par = 200_037
    data = orc.fill(192 * n_par, 0xBA5F)
    want = np.frombuffer(orc.bashF_batch(data), dtype=np.uint8)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    ok = {}
    for v in variants:
        assert tune(0, v) == 0
        if v in ABLATION:
            ok[v] = True
            continue
        for n in (n_par, 1, 63, 64, 65, 4097):
            t = src[: 192 * n].clone()
            eng.bashF_batch_dev(t)
            torch.cuda.synchronize()
            good = np.array_equal(t.cpu().numpy(), want[: 192 * n])
            ok[v] = ok.get(v, True) and good
        print(f"parity v{v}: {'ok' if ok[v] else 'MISMATCH'}", flush=True)
    for logn in (20, 22):
        n = 1 << logn
        st = torch.empty(192 * n, dtype=torch.uint8, device="cuda")
        st.random_(0, 256)
        res = {v: [] for v in variants}
        ghz = {v: [] for v in variants}
        side = torch.cuda.Stream()
        probe = torch.zeros(2, dtype=torch.int64, device="cuda")
        for rnd in range(int(os.environ.get('AB_ROUNDS', '4'))):
            for v in variants:
                tune(0, v)
                us = eng.time_kernel(0, 10, st, n=n) * 1e3
                # clock probe beside the timed launches: spins for ~80 % of their expected duration
                torch.cuda.synchronize()
                eng.lib.bee2hip_internal_clock_probe(ctypes.c_void_p(probe.data_ptr()), ctypes.c_uint(int(us * 60 * 0.8)),
                                                     ctypes.c_void_p(side.cuda_stream))
                res[v].append(eng.time_kernel(0, 60, st, n=n) * 1e3)
                torch.cuda.synchronize()
                c = probe.cpu().numpy()
                ghz[v].append(c[0] / (c[1] * 10.0))
        print(f"--- n = 2^{logn} states, us per launch (4 alternating rounds of 60 launches), G perm/s from the best")
        for v in variants:
            us = res[v]
            med = sorted(us)[len(us) // 2]
            print(f"v{v} {NAMES[v]:<52s} " + " ".join(f"{x:7.1f}" for x in us) +
                  f"   median {med:6.1f}  best {n / min(us) / 1e3:6.2f} G/s  {sum(ghz[v]) / len(ghz[v]):.3f} GHz  {'ok' if ok[v] else 'WRONG'}")
    tune(0, -1)


if __name__ == "__main__":
    main()
