"""Shader clock sustained under each hot kernel (one-wavefront probe on a side stream, s_memtime against the 100 MHz
s_memrealtime): python tools/ab/clock_under_load.py   (on the GPU)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd, goldenlib
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
side = torch.cuda.Stream()
probe = torch.zeros(2, dtype=torch.int64, device="cuda")


def clock(name, fn, ms_est, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    eng.lib.bee2hip_internal_clock_probe(ctypes.c_void_p(probe.data_ptr()), ctypes.c_uint(int(ms_est * 1e3 * reps * 0.8)),
                                         ctypes.c_void_p(side.cuda_stream))
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    c = probe.cpu().numpy()
    print(f"{name:<44s} {c[0] / (c[1] * 10.0):.2f} GHz")


eng.lib.bee2hip_internal_clock_probe(ctypes.c_void_p(probe.data_ptr()), ctypes.c_uint(3000), ctypes.c_void_p(side.cuda_stream))
torch.cuda.synchronize(); c = probe.cpu().numpy(); print(f"{'idle (probe alone)':<44s} {c[0] / (c[1] * 10.0):.2f} GHz")
st = torch.empty(192 << 20, dtype=torch.uint8, device="cuda"); st.random_(0, 256)
clock("bashF, 2^20 states", lambda: eng.bashF_batch_dev(st), 0.09, 200)
kw, c0 = eng.beltCTRStart(bytes(range(32)), bytes(16))
buf = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
clock("beltCTR, 4 GiB", lambda: eng.beltCTR_blocks_dev(buf, kw, c0), 5.0, 8)
del buf
hs, ss, ps = G.bign_base_arrays()
k = (1 << 18) // 2048
dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
codes = torch.empty(1 << 18, dtype=torch.int32, device="cuda")
clock("bign128Verify, 2^18 signatures", lambda: eng.bign128Verify_batch_dev(dh, ds, dp, codes), 2.5, 16)
n13 = 1 << 13
clock("bign128Verify, 2^13 signatures (quads)", lambda: eng.bign128Verify_batch_dev(dh[: 32 * n13], ds[: 48 * n13], dp[: 64 * n13], codes[:n13]), 0.42, 60)
