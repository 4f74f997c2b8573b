"""A/B of the parts of a big verification batch (bign_kernels.hip launch_bign_verify_t: part p + 1 starts when part p's prep
is through, so prep / inv / tail run beside another part's main kernel): ms per batch, one process, alternating.
python tools/ab/verify_split_ab.py   (on the GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd, goldenlib
from bee2_amd.engine import LEVEL_OID
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
tune = eng.lib.bee2hip_internal_tune


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


hs, ss, ps = G.bign_base_arrays()
for e in (17, 18, 19, 20):
    k = (1 << e) // 2048
    dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
    n = 2048 * k
    codes = torch.empty(n, dtype=torch.int32, device="cuda")
    res = {v: [] for v in (1, 2, 3, 4)}
    for rnd in range(3):
        for v in res:
            tune(8, v)
            res[v].append(timed(lambda: eng.bign128Verify_batch_dev(dh, ds, dp, codes), 10))
            assert int((codes != 0).sum()) == 0
    print(f"256-bit curve, 2^{e} signatures: " + "  ".join(f"{v} part(s) {min(r):.3f} ms ({n / min(r) / 1e3:.1f} M/s)" for v, r in res.items()), flush=True)
for l in (192, 256):
    base = G.bign_big[str(l)]["base"]
    n = 1 << 18
    reps = n // len(base) + 1
    h, s, kk = (torch.tensor(list((b"".join(bytes.fromhex(t[f]) for t in base) * reps)[: w * n]), dtype=torch.uint8).cuda()
                for f, w in (("hash", l // 4), ("sig", 3 * l // 8), ("pubkey", l // 2)))
    codes = torch.empty(n, dtype=torch.int32, device="cuda")
    res = {v: [] for v in (1, 2, 4)}
    for rnd in range(2):
        for v in res:
            tune(8, v)
            res[v].append(timed(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], h, s, kk, codes), 4))
            assert int((codes != 0).sum()) == 0
    print(f"{2 * l}-bit curve, 2^18 signatures: " + "  ".join(f"{v} part(s) {min(r):.3f} ms ({n / min(r) / 1e3:.1f} M/s)" for v, r in res.items()), flush=True)
tune(8, 0)
