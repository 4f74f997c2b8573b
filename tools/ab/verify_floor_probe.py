#!/usr/bin/env python3
"""One verification form at one batch size, `reps` times (experiments build): the workload behind the per-kernel times of
profiles/r05_verify_floor.txt (run under rocprofv3 --kernel-trace --stats).  usage: verify_floor_probe.py <log2 n> <form> [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import goldenlib  # noqa: E402

e, form = int(sys.argv[1]), int(sys.argv[2], 0)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
eng = bee2_amd.load_experiments(); eng.set_device(0)
hs, ss, ps = goldenlib.Golden().bign_base_arrays()
n = 1 << e
k = max(1, n // 2048)
dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
codes = torch.empty(n, dtype=torch.int32, device="cuda")
eng.lib.bee2hip_internal_tune(2, form)
for _ in range(5):
    eng.bign128Verify_batch_dev(dh[: 32 * n], ds[: 48 * n], dp[: 64 * n], codes)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng.bign128Verify_batch_dev(dh[: 32 * n], ds[: 48 * n], dp[: 64 * n], codes)
e1.record(); torch.cuda.synchronize()
assert int(codes.count_nonzero().item()) == 0
print(f"n = 2^{e} form {form:#04x}: {e0.elapsed_time(e1) / reps:.4f} ms per batch")
