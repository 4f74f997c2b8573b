"""workgroup size of the signing side's hashing kernels (bign_sign_nonce_kernel / bign_sign_tail_kernel: one 64 KiB belt table per
workgroup + a message row per lane): 256 lanes (round 2) against up to 1024 (bee2hip_internal_tune 12), alternating; device-resident
bignSign2 batches, hipEvents around 10 launches, best of 5.  python tools/ab/sign_wg_ab.py   (on the GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import bee2_amd
from bee2_amd import engine as E
eng = bee2_amd.load_experiments(); eng.set_device(0)
L = eng.lib
rng = np.random.default_rng(5)
for l in (128, 192, 256):
    no, sg = l // 4, 3 * l // 8
    oid = E.LEVEL_OID[l]
    for e in (16, 18):
        n = 1 << e
        pr = rng.integers(0, 256, no * n, dtype=np.uint8); pr[no - 1::no] &= 0x7F
        privs = torch.from_numpy(pr).cuda()
        hashes = torch.from_numpy(rng.integers(0, 256, no * n, dtype=np.uint8)).cuda()
        ts = torch.from_numpy(rng.integers(0, 256, 16 * n, dtype=np.uint8)).cuda()
        sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
        c1 = torch.empty(n, dtype=torch.int32, device="cuda")
        for t_len in (0, 16):
            res, keep = {}, []
            for rnd in range(2):
                for wg in (256, 512, 1024):
                    L.bee2hip_internal_tune(12, wg)
                    kw = dict(t=ts, t_len=16, t_shared=False) if t_len else {}
                    best = 1e9
                    for _ in range(5):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c1, **kw)
                        a.record()
                        for _ in range(10):
                            eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c1, **kw)
                        b.record(); torch.cuda.synchronize()
                        best = min(best, a.elapsed_time(b) / 10)
                    res[wg] = min(res.get(wg, 1e9), best)
                    keep.append(sigs.clone())
            assert all(torch.equal(keep[0], k) for k in keep) and int(c1.abs().sum()) == 0
            print(f"l = {l}  2^{e} signatures, t of {t_len:2d} octets: " + "  ".join(f"wg<={wg}: {res[wg]:7.3f} ms" for wg in (256, 512, 1024)), flush=True)
L.bee2hip_internal_tune(12, 0)
