"""ramped chunk sizes at the ends of the duplex host pipeline (staging.hpp duplex_inplace, bee2hip_internal_tune 9), off / on
alternating: bee2hip_bashF_batch on 2^20 states and the one-shot beltCTR on 1 GiB, host pointers to pageable memory.
python tools/ab/duplex_ramp_ab.py   (on the GPU)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bee2_amd
eng = bee2_amd.load_experiments(); eng.set_device(0)
L = eng.lib
H = eng.beltH()


def clock(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best


n = 1 << 20
st = np.random.default_rng(1).integers(0, 256, 192 * n, dtype=np.uint8)
hn = 1 << 30
buf = np.zeros(hn, dtype=np.uint8)
key, iv = bytes(H[128:160]), bytes(H[192:208])
p = ctypes.c_void_p(buf.ctypes.data)
for rnd in range(3):
    for ramp in (0, 1):
        L.bee2hip_internal_tune(9, ramp)
        for lg in (16, 17):
            L.bee2hip_internal_tune(6, lg)
            dt = clock(lambda: L.bee2hip_bashF_batch(ctypes.c_void_p(st.ctypes.data), ctypes.c_size_t(n)))
            print(f"ramp {ramp}  bashF 2^20 states chunk 2^{lg}: {dt * 1e3:6.2f} ms  {n / dt / 1e6:6.1f} M perm/s", flush=True)
        for lg in (20, 21):
            L.bee2hip_internal_tune(7, lg)
            dt = clock(lambda: L.beltCTR(p, p, ctypes.c_size_t(hn), key, ctypes.c_size_t(32), iv), reps=3)
            print(f"ramp {ramp}  beltCTR 1 GiB chunk 2^{lg}: {dt * 1e3:6.2f} ms  {1 / dt:5.1f} GiB/s", flush=True)
L.bee2hip_internal_tune(6, 16); L.bee2hip_internal_tune(7, 20); L.bee2hip_internal_tune(9, 0)
