import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, torch
import bee2_amd
eng = bee2_amd.load(); eng.set_device(0)
def run(n, length, alg=128, reps=5):
    offs = torch.from_numpy(np.arange(n + 1, dtype=np.int64) * length).cuda()
    data = torch.empty(n * length + 16, dtype=torch.uint8, device="cuda"); data.random_(0, 255)
    dig = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    eng.hash_ragged_dev(alg, data, offs, dig, n); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.hash_ragged_dev(alg, data, offs, dig, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    perms = length // 128 + 1
    print(f"n={n:6d} len={length:7d}: {dt*1e3:8.3f} ms  {n*length/dt/2**30:8.2f} GiB/s  {dt/perms*1e6:6.2f} us per permutation step")
run(1, 256 * 1024); run(8, 256 * 1024); run(64, 256 * 1024); run(1024, 256 * 1024); run(8192, 64 * 1024); run(20000, 64 * 1024); run(1 << 17, 8192)
run(1, 4000); run(1 << 16, 4000)
