import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")]
import numpy as np, torch
import bee2_amd
eng = bee2_amd.load(); eng.set_device(0)
def run(n, length, alg=0, reps=5):
    offs = torch.from_numpy(np.arange(n + 1, dtype=np.int64) * length).cuda()
    data = torch.empty(n * length + 16, dtype=torch.uint8, device="cuda"); data.random_(0, 255)
    dig = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
    eng.hash_ragged_dev(alg, data, offs, dig, n); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.hash_ragged_dev(alg, data, offs, dig, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"alg={alg} n={n:7d} len={length:6d}: {dt*1e3:8.3f} ms  {n*length/dt/2**30:8.2f} GiB/s")
for alg in (0, 128, 256):
    run(1 << 18, 1000, alg); run(1 << 16, 4000, alg); run(1 << 20, 256, alg); run(1 << 14, 65536, alg)
