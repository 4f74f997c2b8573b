#!/usr/bin/env python3
"""Verification kernels with the multiply-adds of a column in pairs (VtOpsP, bign_dev.hpp mac2): ms per device-resident batch by
curve, batch size and which kernels take the paired form -- experiments build, bee2hip_internal_tune(19, v): v = 0 never, 1 always, -1 = the product's choice.  usage: python tools/ab/verify_pairs_ab.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import goldenlib  # noqa: E402
from bee2_amd.engine import LEVEL_OID  # noqa: E402

eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
tune = eng.lib.bee2hip_internal_tune


def t(fn, reps):
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


print("ms per batch: main kernel as round 3 | main kernel with paired multiply-adds | the product's choice")
for l in (128, 192, 256):
    if l == 128:
        hs, ss, ps = G.bign_base_arrays()
        k = (1 << 19) // 2048
        th, ts, tp = (torch.from_numpy(np.frombuffer(x * k, dtype=np.uint8).copy()).cuda() for x in (hs, ss, ps))
        sizes = (16, 17, 18, 19)
    else:
        base = G.bign_big[str(l)]["base"]
        reps_l = (1 << 18) // len(base) + 1
        th, ts, tp = (torch.from_numpy(np.frombuffer(b"".join(bytes.fromhex(x[key]) for x in base) * reps_l, dtype=np.uint8).copy()).cuda()
                      for key in ("hash", "sig", "pubkey"))
        sizes = (16, 17, 18)
    no = l // 4
    for e in sizes:
        m = 1 << e
        tc = torch.empty(m, dtype=torch.int32, device="cuda")
        args = (th[: no * m], ts[: (no + no // 2) * m], tp[: 2 * no * m], tc)
        row = []
        for v in (0, 1, -1):
            tune(19, v)
            if l == 128 and v != -1:
                tune(2, 1)                       # the 32-bit kernels at every size (2^16 would take the 29-bit main kernel)
            row.append(t(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], *args), 6))
            tune(2, 0)
            assert int(tc.count_nonzero().item()) == 0
        tune(19, -1)
        print(f"bign-curve{2 * l}v1 2^{e}: " + "  ".join(f"{x:8.3f}" for x in row))
