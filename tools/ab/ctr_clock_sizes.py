"""beltCTR at 256 MiB / 1 GiB / 4 GiB / 16 GiB in steady state (0.5 s of back-to-back launches first): ms per launch,
GiB/s, and the shader clock a one-wavefront probe on a side stream sees meanwhile -> cycles per block.
python tools/ab/ctr_clock_sizes.py [variant]   (on the GPU)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bee2_amd
eng = bee2_amd.load_experiments(); eng.set_device(0)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng.lib.bee2hip_internal_tune(1, v)
side = torch.cuda.Stream()
probe = torch.zeros(2, dtype=torch.int64, device="cuda")
kw, c0 = eng.beltCTRStart(bytes(range(32)), bytes(16))
for logn in (24, 26, 28, 30):
    n = 1 << logn
    st = torch.empty(16 * n, dtype=torch.uint8, device="cuda"); st.random_(0, 256)
    ms0 = eng.time_kernel(1, 3, st, n=n)
    reps = max(4, int(500 / ms0))
    eng.time_kernel(1, reps, st, n=n)                       # warm: 0.5 s of load
    torch.cuda.synchronize()
    eng.lib.bee2hip_internal_clock_probe(ctypes.c_void_p(probe.data_ptr()), ctypes.c_uint(int(ms0 * 1e3 * reps * 0.7)),
                                         ctypes.c_void_p(side.cuda_stream))
    ms = eng.time_kernel(1, reps, st, n=n)
    torch.cuda.synchronize()
    c = probe.cpu().numpy()
    ghz = c[0] / (c[1] * 10.0)
    print(f"variant {v}: {16 * n / 2**30:6.2f} GiB x {reps:4d}: {ms:8.3f} ms/launch {16 * n / ms / 2**30 * 1e3:7.1f} GiB/s  clock {ghz:.3f} GHz  "
          f"-> {ms * 1e-3 * ghz * 1e9 * 256 / n:7.2f} CU-cycles per block", flush=True)
    del st
