#!/usr/bin/env python3
"""A/B of the lanes of bign_inv_kernel (signatures per shared inversion = n / lanes, at most 16): experiments build, tune 23.  The one-signer
entry (where the inversion kernel is 15 % of a batch) and the general entry at 2^17 .. 2^20 signatures.  usage: python tools/ab/inv_lanes_ab.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
from bee2_amd.engine import LEVEL_OID  # noqa: E402

eng = bee2_amd.load_experiments(); eng.set_device(0)
tune = eng.lib.bee2hip_internal_tune


def t(fn, reps=10):
    for _ in range(2):
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


for l in (128, 192, 256):
    no, sg = l // 4, 3 * l // 8
    nmax = 1 << (20 if l == 128 else 18)
    g = torch.Generator(device="cuda"); g.manual_seed(l)
    h = torch.empty(no * nmax, dtype=torch.uint8, device="cuda"); h.view(torch.int64).random_(generator=g)
    d = bytes((l + 3 * i) & 255 for i in range(no - 1)) + b"\x21"
    p = torch.empty(2 * no, dtype=torch.uint8, device="cuda"); c1 = torch.empty(1, dtype=torch.int32, device="cuda")
    eng.bignPubkeyCalcL_batch_dev(l, torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).cuda(), p, c1)
    s = torch.empty(sg * nmax, dtype=torch.uint8, device="cuda"); cs = torch.empty(nmax, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(l, LEVEL_OID[l], h, torch.from_numpy(np.frombuffer(d * nmax, dtype=np.uint8).copy()).cuda(), s, cs)
    torch.cuda.synchronize()
    pub = p.cpu().numpy().tobytes()
    keys = p.repeat(nmax)
    codes = torch.empty(nmax, dtype=torch.int32, device="cuda")
    tune(20, 0)
    for e in (17, 18, 19, 20):
        n = 1 << e
        if n > nmax:
            break
        line = f"l = {l} 2^{e}:"
        for lg in (0, 14, 15, 16, 17, 18):
            tune(23, lg)
            m1 = t(lambda: eng.bignVerifyL_onekey_batch_dev(l, LEVEL_OID[l], h[: no * n], s[: sg * n], pub, codes[:n]))
            assert not codes[:n].any()
            mg = t(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], h[: no * n], s[: sg * n], keys[: 2 * no * n], codes[:n]), reps=4)
            assert not codes[:n].any()
            line += f"  lanes 2^{lg if lg else '(15|16)'}: {m1:.3f} | {mg:.3f}"
        tune(23, 0)
        print(line + "   (ms: one signer | general)")
    tune(20, -1)
