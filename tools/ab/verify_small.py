"""bign128Verify on a small device-resident batch, a few times (for rocprofv3 --kernel-trace --stats: which kernel
holds the latency floor): python tools/ab/verify_small.py <log2 n> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd, goldenlib
e = int(sys.argv[1]) if len(sys.argv) > 1 else 13
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
hs, ss, ps = G.bign_base_arrays()
k = max(1, (1 << e) // 2048)
dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
n = min(1 << e, 2048 * k)
codes = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    eng.time_kernel(2, 3, dh, ds, dp, codes, n=n)
ms = eng.time_kernel(2, reps, dh, ds, dp, codes, n=n)
print(f"2^{e} signatures: {ms:.3f} ms per batch, {n / ms / 1e3:.2f} M verifies/s")
