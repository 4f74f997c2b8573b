"""Host-pointer calls through the pinned (zero-copy) staging buffer against device staging + hipMemcpy
(bee2hip_internal_tune(3, limit)): us per call for sizes around the switch.  Run on the GPU."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd
eng = bee2_amd.load_experiments(); eng.set_device(0)
tune = eng.lib.bee2hip_internal_tune
key = bytes(range(32)); iv = bytes(16)


def bench(fn, reps=200):
    for _ in range(20):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e6


print("us per call:              size   pinned   hipMemcpy")
for size in (16, 192, 1024, 4096, 16384, 32768, 65536):
    msg = os.urandom(size)
    row = {}
    for name, limit in (("pinned", 65536), ("copy", 0)):
        tune(3, limit)
        row[name] = (bench(lambda: eng.beltHash(msg)), bench(lambda: eng.bashHash(128, msg)), bench(lambda: eng.beltCTR(msg, key, iv)),
                     bench(lambda: eng.beltMAC(msg, key)))
    for i, what in enumerate(("beltHash", "bashHash(128)", "beltCTR", "beltMAC")):
        print(f"{what:<16s} {size:>12d} {row['pinned'][i]:8.1f} {row['copy'][i]:8.1f}")
tune(3, 65536)
