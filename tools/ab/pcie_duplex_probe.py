"""Feasibility probe for a chunked duplex host pipeline: can H2D and D2H of PAGEABLE host memory
overlap when issued from two host threads on two streams?  (Run on the GPU box.)"""
import threading, time
import numpy as np, torch
N = 1 << 30
h_in = np.ones(N, dtype=np.uint8); h_out = np.empty(N, dtype=np.uint8)
d_a = torch.empty(N, dtype=torch.uint8, device="cuda"); d_b = torch.ones(N, dtype=torch.uint8, device="cuda")
t_in = torch.from_numpy(h_in); t_out = torch.from_numpy(h_out)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(t_in, non_blocking=True); s1.synchronize()
def d2h():
    with torch.cuda.stream(s2):
        t_out.copy_(d_b, non_blocking=True); s2.synchronize()
def clock(fns, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        best = min(best, time.perf_counter() - t0)
    return best
h2d(); d2h()
a, b, c = clock([h2d]), clock([d2h]), clock([h2d, d2h])
print(f"pageable 1 GiB: H2D {1/a:.1f} GiB/s, D2H {1/b:.1f} GiB/s, both concurrently {c*1e3:.1f} ms "
      f"(sum of singles {1e3*(a+b):.1f} ms, max {1e3*max(a,b):.1f} ms) -> overlap factor {(a+b)/c:.2f}")
p_in = torch.ones(N, dtype=torch.uint8).pin_memory(); p_out = torch.empty(N, dtype=torch.uint8).pin_memory()
def h2dp():
    with torch.cuda.stream(s1):
        d_a.copy_(p_in, non_blocking=True); s1.synchronize()
def d2hp():
    with torch.cuda.stream(s2):
        p_out.copy_(d_b, non_blocking=True); s2.synchronize()
h2dp(); d2hp()
a, b, c = clock([h2dp]), clock([d2hp]), clock([h2dp, d2hp])
print(f"pinned   1 GiB: H2D {1/a:.1f} GiB/s, D2H {1/b:.1f} GiB/s, both concurrently {c*1e3:.1f} ms -> overlap factor {(a+b)/c:.2f}")
t0 = time.perf_counter(); x = torch.ones(N, dtype=torch.uint8); torch.cuda.cudart().cudaHostRegister(x.data_ptr(), N, 0); dt = time.perf_counter() - t0
print(f"hipHostRegister of 1 GiB (incl. first touch): {dt*1e3:.1f} ms")
