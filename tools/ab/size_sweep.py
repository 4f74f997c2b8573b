"""Throughput vs batch size for the three kernels (device-resident, hipEvent-timed launches).
Run on the GPU box: python tools/ab/size_sweep.py > gpurun_out/size_sweep.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd, goldenlib
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden(); H = eng.beltH()
print("bashF: states  us/launch  Gperm/s  TB/s")
for e in (12, 14, 16, 17, 18, 19, 20, 22, 24):
    n = 1 << e
    st = torch.empty(192 * n, dtype=torch.uint8, device="cuda"); st.view(torch.int64).random_()
    for _ in range(5): eng.time_kernel(0, 20, st, n=n)
    ms = eng.time_kernel(0, max(20, min(2000, (1 << 26) // n)), st, n=n)
    print(f"2^{e:<3} {ms*1e3:10.1f} {n/ms/1e6:8.2f} {384*n/ms/1e9:6.2f}")
    del st
print("beltCTR: bytes  ms/launch  GiB/s")
kw, c0 = eng.beltCTRStart(H[128:160], H[192:208])
for e in range(20, 35, 2):
    nb = (1 << e) // 16
    buf = torch.empty(16 * nb, dtype=torch.uint8, device="cuda")
    for _ in range(3): eng.time_kernel(1, 5, buf, n=nb)
    ms = eng.time_kernel(1, max(3, min(200, (1 << 33) >> e)), buf, n=nb)
    print(f"2^{e:<3} {ms:10.3f} {(1 << e)/2**30/(ms*1e-3):8.1f}")
    del buf
print("bign128Verify: sigs  ms/batch  Mverify/s")
hs, ss, ps = G.bign_base_arrays()
for e in range(11, 21):
    reps = (1 << e) // 2048
    dh, ds, dp = (torch.from_numpy(np.frombuffer(x * reps, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
    n = 2048 * reps
    codes = torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(3): eng.time_kernel(2, 3, dh, ds, dp, codes, n=n)
    ms = eng.time_kernel(2, max(3, min(100, (1 << 22) >> e)), dh, ds, dp, codes, n=n)
    print(f"2^{e:<3} {ms:10.3f} {n/ms/1e3:8.2f}")
