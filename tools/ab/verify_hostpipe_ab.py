"""bee2hip_bignVerify_batch with HOST pointers: one upload then the kernels, against chunks of 2^18 uploaded while the previous
chunk is verified (capi.hip, bee2hip_internal_tune 11), alternating.  python tools/ab/verify_hostpipe_ab.py   (on the GPU)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bee2_amd, goldenlib
eng = bee2_amd.load_experiments(); eng.set_device(0)
L = eng.lib
G = goldenlib.Golden()
hs, ss, ps = G.bign_base_arrays()
P = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
oid = bytes.fromhex("06092A7000020022651F51")
for e in (18, 19, 20, 21):
    k = (1 << e) // 2048
    h, s, p = (np.frombuffer(x * k, dtype=np.uint8).copy() for x in (hs, ss, ps))
    n = 2048 * k
    codes = (ctypes.c_uint32 * n)()
    def call():
        rc = L.bee2hip_bignVerify_batch(ctypes.byref(P), oid, ctypes.c_size_t(len(oid)), ctypes.c_void_p(h.ctypes.data),
                                        ctypes.c_void_p(s.ctypes.data), ctypes.c_void_p(p.ctypes.data), ctypes.c_size_t(n), codes)
        assert rc == 0
    row = []
    for rnd in range(3):
        for knob in (0, 1):
            L.bee2hip_internal_tune(11, knob)
            call()
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
            row.append((knob, best))
    assert not any(codes)
    one = min(t for k_, t in row if k_ == 0); pipe = min(t for k_, t in row if k_ == 1)
    print(f"2^{e} signatures, host pointers: one piece {one * 1e3:7.2f} ms ({n / one / 1e6:6.1f} M/s)   chunked {pipe * 1e3:7.2f} ms ({n / pipe / 1e6:6.1f} M/s)", flush=True)
L.bee2hip_internal_tune(11, 1)
