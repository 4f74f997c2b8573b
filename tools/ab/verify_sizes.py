#!/usr/bin/env python3
"""ms per device-resident verification batch by size, all three curves, product entry points (BEE2HIP_LIB picks the build: run once per
build inside ONE gpurun call for an A/B).  usage: python tools/ab/verify_sizes.py [tag]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import goldenlib  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
eng = bee2_amd.load(); eng.set_device(0)
G = goldenlib.Golden()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


hs, ss, ps = G.bign_base_arrays()
out = []
for e in (10, 12, 13, 14, 15, 16, 17, 18, 19):
    k = max(1, (1 << e) // 2048)
    dh, ds, dp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
    n = min(1 << e, 2048 * k)
    codes = torch.empty(n, dtype=torch.int32, device="cuda")
    ms = t(lambda: eng.bign128Verify_batch_dev(dh[: 32 * n], ds[: 48 * n], dp[: 64 * n], codes))
    assert int(codes.count_nonzero().item()) == 0
    out.append(f"2^{e} {ms:.3f}")
print(f"{tag:6s} bign-curve256v1, ms per batch: " + "  ".join(out))
from bee2_amd.engine import LEVEL_OID  # noqa: E402
for l in (192, 256):
    base = G.bign_big[str(l)]["base"]
    reps_l = (1 << 18) // len(base) + 1
    th, ts, tp = (torch.from_numpy(np.frombuffer(b"".join(bytes.fromhex(x[k]) for x in base) * reps_l, dtype=np.uint8).copy()).cuda()
                  for k in ("hash", "sig", "pubkey"))
    no = l // 4
    out = []
    for e in (10, 13, 14, 15, 16, 17, 18):
        m = 1 << e
        tc = torch.empty(m, dtype=torch.int32, device="cuda")
        args = (th[: no * m], ts[: (no + no // 2) * m], tp[: 2 * no * m], tc)
        ms = t(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], *args), reps=8)
        assert int(tc.count_nonzero().item()) == 0
        out.append(f"2^{e} {ms:.3f}")
    print(f"{tag:6s} bign-curve{2 * l}v1, ms per batch: " + "  ".join(out))
