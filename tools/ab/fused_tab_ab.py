"""A/B of the belt table of the fused bash512 + beltMAC kernel inside one process on one box (run on the GPU:
python tools/ab/fused_tab_ab.py): tune 13 = 1 the four-table 128 KiB layout (rounds 1-2), 2 the two-table layout with
one-instruction LDS addresses (BeltTabTwoP), 0 the product.  Each form is first held to the oracle on 4096 messages."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import orclib  # noqa: E402


def main():
    eng = bee2_amd.load_experiments()
    eng.set_device(0)
    orc = orclib.load()
    tune = eng.lib.bee2hip_internal_tune
    key = bytes(range(32))
    ml = 4096
    n0 = 4096
    data = orc.fill(ml * n0, 0x4D1C)
    dig, tag = orc.mixed_batch(data, ml, key, nthreads=16)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for v in (1, 2, 0):
        tune(13, v)
        d = torch.zeros(64 * n0, dtype=torch.uint8, device="cuda")
        t = torch.zeros(8 * n0, dtype=torch.uint8, device="cuda")
        eng.bashHash_beltMAC_batch_dev(src, ml, 256, key, d, t, n=n0)
        torch.cuda.synchronize()
        ok = d.cpu().numpy().tobytes() == dig and t.cpu().numpy().tobytes() == tag
        print(f"parity tab {v}: {'ok' if ok else 'MISMATCH'}", flush=True)
    n = 1 << 21
    msgs = torch.empty(ml * n, dtype=torch.uint8, device="cuda")
    msgs.random_(0, 256)
    d = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    t = torch.empty(8 * n, dtype=torch.uint8, device="cuda")
    res = {1: [], 2: []}
    for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
        for v in (1, 2):
            tune(13, v)
            eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, d, t, n=n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, d, t, n=n)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 3)
    for v in (1, 2):
        r = sorted(res[v])
        print(f"tab {v}: {r[0]:8.3f} ms min, {r[len(r) // 2]:8.3f} median  {n / r[0] / 1e3:7.2f} M messages/s")
    tune(13, 0)


if __name__ == "__main__":
    main()
