import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch, bee2_amd
eng = bee2_amd.load_experiments(); eng.set_device(0)
n = 1 << 20
st = torch.empty(192*n, dtype=torch.uint8, device="cuda"); st.view(torch.int64).random_()
for _ in range(50): eng.bashF_batch_dev(st)
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K): eng.bashF_batch_dev(st)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_py = time.perf_counter() - t0
t0 = time.perf_counter(); ms = eng.time_kernel(0, K, st, n=n); t_c = time.perf_counter() - t0
# raw ctypes loop without python helpers
f = eng.lib.bee2hip_bashF_batch_dev; p = ctypes.c_void_p(st.data_ptr()); nn = ctypes.c_size_t(n)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): f(p, nn, None)
t_issue2 = time.perf_counter() - t0
torch.cuda.synchronize(); t_raw = time.perf_counter() - t0
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sp = ctypes.c_void_p(s.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): f(p, nn, sp)
    torch.cuda.synchronize(); t_ns = time.perf_counter() - t0
print(f"python helper loop: {t_py/K*1e6:.1f} us/step (issue {t_issue/K*1e6:.1f}); raw ctypes null stream: {t_raw/K*1e6:.1f} (issue {t_issue2/K*1e6:.1f}); raw non-null stream: {t_ns/K*1e6:.1f}; C loop wall {t_c/K*1e6:.1f}, events {ms*1e3:.1f}")
