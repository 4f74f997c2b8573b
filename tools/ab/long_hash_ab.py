#!/usr/bin/env python3
"""A/B of belt_hash_long_kernel's table / workgroup form on the GPU box (experiments build, tune 16): one 256 KiB chain alone,
many long chains, and bench.py's ragged distribution.  usage: python tools/ab/long_hash_ab.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402

eng = bee2_amd.load_experiments(); eng.set_device(0)


def run(lens, form, reps=3, fork=1):
    lens = np.asarray(lens, dtype=np.int64)
    offs = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    data = torch.zeros(int(offs[-1]) // 8 * 8 + 16, dtype=torch.uint8, device="cuda")
    torch.manual_seed(1234)
    data.view(torch.int64).random_()
    doff = torch.from_numpy(offs).cuda()
    order = torch.from_numpy(np.argsort(-lens, kind="stable").astype(np.int32)).cuda()
    dig = torch.empty(32 * len(lens), dtype=torch.uint8, device="cuda")
    eng.lib.bee2hip_internal_tune(16, form)
    eng.lib.bee2hip_internal_tune(17, fork)
    best, ref = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.hash_ragged_dev(0, data, doff, dig, len(lens), order=order)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    eng.lib.bee2hip_internal_tune(16, 0)
    eng.lib.bee2hip_internal_tune(17, 1)
    return best * 1e3, dig.cpu().numpy().tobytes()


rng = np.random.default_rng(0x4D1C)
cases = {"65536 x 1000 B + one 256 KiB": [1 << 18] + [1000] * 65535, "131072 x 8 KiB (beyond 2^16 messages: product = pair form)": [1 << 13] * (1 << 17), "one 256 KiB message": [1 << 18], "64 x 256 KiB": [1 << 18] * 64, "4096 x 256 KiB": [1 << 18] * 4096,
         "4096 x 16 KiB": [1 << 14] * 4096,
         "bench ragged (65536, log-uniform < 256 KiB)": (np.floor(2.0 ** (18.0 * rng.random(1 << 16))).astype(np.int64) - 1).tolist()}
print("ms per batch: pair per message, 4 KiB table, ONE queue (r03 product) | EIGHT lanes (each G-box shared by a quad: one byte per lane), one queue | "
      "eight lanes, long chains and short messages on TWO queues = product up to 2^16 messages")
for name, lens in cases.items():
    out = [run(lens, 1, fork=0), run(lens, 7, fork=0), run(lens, 0, fork=1)]
    assert all(o[1] == out[0][1] for o in out), name
    print(f"{name:48s} " + " ".join(f"{o[0]:9.2f}" for o in out))
