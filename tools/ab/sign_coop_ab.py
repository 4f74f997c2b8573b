"""k G of the signing side by 1 / 4 / 16 / 64 lanes per scalar (bee2hip_internal_tune 10), by batch size:
device-resident bignPubkeyCalc and bignSign2 batches, wall clock around launch + synchronize (min of 7), and the single-call
drop-in latencies.  python tools/ab/sign_coop_ab.py [l]   (on the GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd
from bee2_amd import engine as E
eng = bee2_amd.load_experiments(); eng.set_device(0)
L = eng.lib
l = int(sys.argv[1]) if len(sys.argv) > 1 else 128
no, sg = l // 4, 3 * l // 8
oid = E.LEVEL_OID[l]
rng = np.random.default_rng(3)


def clock(fn, reps=7):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e6


FORMS = (101, 102, 1, 4, 16, 64)
print(f"l = {l}: us per batch, k G by 1 lane (4-bit windows) / 1 lane (signed 6-bit, complete additions) / 1 lane (signed 6-bit, Jacobian) / 4 / 16 / 64 lanes per scalar")
for e in (0, 14, 16, 17, 18):
    n = 1 << e
    pr = rng.integers(0, 256, no * n, dtype=np.uint8); pr[no - 1::no] &= 0x7F
    privs = torch.from_numpy(pr).cuda()
    hashes = torch.from_numpy(rng.integers(0, 256, no * n, dtype=np.uint8)).cuda()
    pubs = torch.empty(2 * no * n, dtype=torch.uint8, device="cuda")
    sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
    c1 = torch.empty(n, dtype=torch.int32, device="cuda")
    row, keep = [], []
    for v in FORMS:
        if 1 < v < 100 and n * v > (1 << 21):
            row.append(None); continue
        L.bee2hip_internal_tune(10, v)
        tp = clock(lambda: eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, c1))
        ts = clock(lambda: eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c1))
        torch.cuda.synchronize()
        keep.append((pubs.clone(), sigs.clone()))
        row.append((tp, ts))
    assert all(torch.equal(keep[0][0], k[0]) and torch.equal(keep[0][1], k[1]) for k in keep)
    L.bee2hip_internal_tune(10, 0)
    auto = clock(lambda: eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c1))
    fmt = lambda i: " / ".join("      --" if r is None else f"{r[i]:8.1f}" for r in row)
    print(f"  n = 2^{e:<2d}  pubkey calc {fmt(0)}    sign2 {fmt(1)}    product sign2 {auto:8.1f}", flush=True)

P = eng.bignParamsStd(E.CURVE_NAME[l])
priv = bytes(pr[:no]); h = bytes(range(no))
for v in (1, 0):
    L.bee2hip_internal_tune(10, v)
    tc = clock(lambda: eng.bignPubkeyCalc(P, priv), reps=20)
    ts = clock(lambda: eng.bignSign2(P, oid, h, priv, None), reps=20)
    print(f"  drop-in single call, {'product (64 lanes)' if v == 0 else 'one lane'} per scalar: bignPubkeyCalc {tc:7.1f} us   bignSign2 {ts:7.1f} us")
L.bee2hip_internal_tune(10, 0)
