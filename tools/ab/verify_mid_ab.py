#!/usr/bin/env python3
"""A/B of the verification forms between 2^15 and 2^17 signatures on the 256-bit curve (experiments build, bee2hip_internal_tune 2):
0 = the product's dispatch by size (pairs of lanes up to 2^15, the 29-bit one-lane kernel up to 2^16, the 32-bit kernels above),
0x23 = one signature per PAIR of lanes at every size, 2 = the 29-bit one-lane kernel, 1 = the 32-bit kernels.
Device-resident batches, hipEvents, verdicts asserted.  usage: python tools/ab/verify_mid_ab.py [log2 of the largest batch, default 17]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import goldenlib  # noqa: E402

eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
hs, ss, ps = G.bign_base_arrays()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


TOP = int(sys.argv[1]) if len(sys.argv) > 1 else 17
k = (1 << TOP) // 2048
dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
print("ms per batch by form (0 = product dispatch, 0x23 = pairs, 2 = one lane 29-bit, 1 = one lane 32-bit)")
for n in [1 << 13, 1 << 14, 1 << 15, 40960, 49152, 57344, 1 << 16, 81920, 98304, 1 << 17] + [1 << e for e in range(18, TOP + 1)]:
    codes = torch.empty(n, dtype=torch.int32, device="cuda")
    line = f"n = {n:6d}:"
    for form in (0, 0x23, 2, 1):
        eng.lib.bee2hip_internal_tune(2, form)
        ms = t(lambda: eng.bign128Verify_batch_dev(dh[: 32 * n], ds[: 48 * n], dp[: 64 * n], codes))
        assert int(codes.count_nonzero().item()) == 0
        line += f"  form {form:#04x}: {ms:.3f} ms ({n / ms / 1e3:6.1f} M/s)"
    eng.lib.bee2hip_internal_tune(2, 0)
    print(line)
