"""bign192Verify / bign256Verify batches on device-resident data: ms per batch by size, r01 kernels (forced) against the
size-selected ones (quads on 28- / 27-bit limbs up to 2^14 signatures).  Run on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bee2_amd, goldenlib
from bee2_amd.engine import LEVEL_OID
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
tune = eng.lib.bee2hip_internal_tune


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for l in (192, 256):
    base = G.bign_big[str(l)]["base"]
    for n in (1 << 8, 1 << 11, 1 << 13, 1 << 14, 3 << 13, 1 << 15, 5 << 13, 3 << 14, 1 << 16):
        reps = n // len(base) + 1
        h, s, k = (torch.tensor(list((b"".join(bytes.fromhex(t[f]) for t in base) * reps)[: w * n]), dtype=torch.uint8).cuda()
                   for f, w in (("hash", l // 4), ("sig", 3 * l // 8), ("pubkey", l // 2)))
        codes = torch.empty(n, dtype=torch.int32, device="cuda")
        row = []
        for path in (1, 3, 0):
            tune(2, path)
            ms = timed(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], h, s, k, codes))
            assert int((codes != 0).sum()) == 0
            row.append(ms)
        tune(2, 0)
        print(f"l = {l}, {n} signatures: one-lane kernels {row[0]:.3f} ms, quads {row[1]:.3f} ms, by size {row[2]:.3f} ms "
              f"(x{row[0] / row[2]:.2f}); {n / row[2] / 1e3:.2f} M verifies/s")
