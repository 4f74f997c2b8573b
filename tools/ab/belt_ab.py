"""A/B of the beltCTR kernel variants (table layout x blocks per lane) inside ONE process on one box
(run on the GPU: python tools/ab/belt_ab.py [variants...]).  Each variant is first checked against the
product kernel's output (itself pinned to the oracle by tests/test_gpu_belt.py) on a ragged stream,
then timed (hipEvents around `reps` launches, bee2hip_time_kernel) in alternation with the others."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import orclib  # noqa: E402

NAMES = {0: "product: variant 20 (SDWA addresses, v_lshl_or combine, three register sets, hoisted G-box, contiguous ranges, nt accesses)",
         21: "the product before the SDWA addresses: two-table 64 KiB, 1 block/lane, contiguous ranges, nt accesses, round-1 G-box of (c, d) hoisted",
         14: "round 3 without the hoisted round-1 G-box (56 G-boxes per block)",
         16: "as 15 with the post-shifts folded into two v_lshl_or_b32",
         17: "as 15 with the hoisted round-1 G-box",
         18: "as 16 with the hoisted round-1 G-box",
         19: "as 17 with 2 blocks per lane",
         22: "as 20 with all four entries of a G-box awaited at one point (one s_waitcnt per G-box)",
         20: "as 18 with three address-register sets instead of seven",
         15: "as 14 with every LDS address made by one v_mov_b32_sdwa (byte k of x into byte 1 of a register that holds the lane base)",
         13: "round-2 product: two-table 64 KiB, 1 block/lane, tiles dealt round-robin, plain loads/stores",
         1: "two-table, 2 blocks/lane, 8 w/SIMD (45 VGPRs) = the r01 product",
         2: "two-table, 3 blocks/lane, 8 w/SIMD (64 VGPRs, 2 spills)",
         3: "two-table, 4 blocks/lane, 8 w/SIMD (64 VGPRs, 40 spills)",
         4: "four-table 128 KiB (no post-shifts), 2 blocks/lane, 4 w/SIMD (45 VGPRs)",
         5: "four-table, 3 blocks/lane, 4 w/SIMD (66 VGPRs)",
         6: "four-table, 4 blocks/lane, 4 w/SIMD (88 VGPRs)",
         7: "product + non-temporal loads/stores",
         8: "product, one contiguous range of tiles per workgroup",
         9: "product, contiguous ranges + non-temporal",
         10: "hybrid: 8 of 56 G-boxes (32 of 224 lookups) through the vector L1",
         11: "hybrid: 4 of 56 G-boxes (16 lookups) through the vector L1",
         12: "hybrid: 2 of 56 G-boxes (8 lookups) through the vector L1"}


def main():
    variants = [int(x) for x in sys.argv[1:]] or sorted(NAMES)
    eng = bee2_amd.load_experiments()
    eng.set_device(0)
    orc = orclib.load()
    tune = eng.lib.bee2hip_internal_tune
    tune.restype = ctypes.c_uint32
    key = bytes(range(32))
    iv = bytes(range(100, 116))
    kw, c0 = eng.beltCTRStart(key, iv)
    nb = 300_017
    data = orc.fill(16 * nb, 0xC7A)
    want = np.frombuffer(orc.ctr(data, key, iv), dtype=np.uint8)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for v in variants:
        assert tune(1, v) == 0
        good = True
        for n in (nb, 1, 1023, 1024, 1025, 2049, 4097):
            t = src[: 16 * n].clone()
            eng.beltCTR_blocks_dev(t, kw, c0)
            torch.cuda.synchronize()
            good = good and np.array_equal(t.cpu().numpy(), want[: 16 * n])
        print(f"parity v{v}: {'ok' if good else 'MISMATCH'}", flush=True)
    for logn in (26, 30):
        n = 1 << logn                    # blocks: 1 GiB and 16 GiB
        st = torch.empty(16 * n, dtype=torch.uint8, device="cuda")
        st.random_(0, 256)
        res = {v: [] for v in variants}
        for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
            for v in variants:
                tune(1, v)
                res[v].append(eng.time_kernel(1, 5 if logn == 30 else 20, st, n=n))
        print(f"--- 2^{logn} blocks ({16 * n >> 30} GiB), ms per launch (min / median of {len(res[variants[0]])}) and GiB/s at the min")
        for v in variants:
            r = sorted(res[v])
            print(f"v{v:<2d} {r[0]:8.3f} {r[len(r) // 2]:8.3f}  {16 * n / r[0] / 2**30 * 1e3:7.1f} GiB/s   {NAMES.get(v, '')}")
        del st
    tune(1, 0)


if __name__ == "__main__":
    main()
