"""Is the 16 GiB beltCTR rate (800-825 GiB/s) below the 1 GiB rate (875-885) because of memory, or because a
longer run settles at a lower clock?  Same kernel, same 1 GiB buffer, timed over 5 ... 800 back-to-back launches
(6 ms ... 0.9 s of continuous load), then the 16 GiB buffer over 1 ... 20 launches, each after 2 s of idle.
Run on the GPU: python tools/ab/ctr_dvfs_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bee2_amd
eng = bee2_amd.load_experiments(); eng.set_device(0)
for logn, rep_list in ((26, (5, 20, 100, 400, 800)), (30, (1, 2, 5, 20))):
    n = 1 << logn
    st = torch.empty(16 * n, dtype=torch.uint8, device="cuda"); st.random_(0, 256)
    eng.time_kernel(1, 2, st, n=n)
    for reps in rep_list:
        torch.cuda.synchronize(); time.sleep(2.0)
        ms = eng.time_kernel(1, reps, st, n=n)
        print(f"{16 * n >> 30:3d} GiB x {reps:4d} launches after 2 s idle: {ms:8.3f} ms per launch, {16 * n / ms / 2**30 * 1e3:7.1f} GiB/s "
              f"({ms * reps:7.1f} ms of load)", flush=True)
    del st
