"""ms per 2^18-signature verification batch on the 256-bit curve (device-resident, hipEvents; min and median of 5 x 20 launches after
0.5 s of warm-up) -- for A/B of two library builds: BEE2HIP_LIB=<lib> python tools/ab/verify_time.py [log2 n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd, goldenlib
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
e = int(sys.argv[1]) if len(sys.argv) > 1 else 18
hs, ss, ps = G.bign_base_arrays()
k = (1 << e) // 2048
dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
n = 2048 * k
codes = torch.empty(n, dtype=torch.int32, device="cuda")
eng.time_kernel(2, 200, dh, ds, dp, codes, n=n)
r = sorted(eng.time_kernel(2, 20, dh, ds, dp, codes, n=n) for _ in range(5))
assert int((codes != 0).sum()) == 0
print(f"{os.environ.get('BEE2HIP_LIB', 'in-tree')[-40:]:>40s}  2^{e}: min {r[0]:.4f} ms  median {r[2]:.4f} ms  {n / r[0] / 1e3:.1f} M/s")
