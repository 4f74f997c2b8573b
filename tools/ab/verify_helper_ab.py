"""One signature per quad (tune 0x43) against quad + helper quad (0x93 / 0xA3, 64 / 256 lanes per block: the comb of u rides on a second quad) on
device-resident batches, all three levels: ms per batch by size.  Run on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bee2_amd, goldenlib
from bee2_amd.engine import LEVEL_OID
eng = bee2_amd.load_experiments(); eng.set_device(0)
G = goldenlib.Golden()
tune = eng.lib.bee2hip_internal_tune


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for l in (128, 192, 256):
    if l == 128:
        hs, ss, ps = G.bign_base_arrays()
        m = len(hs) // 32
        base = None
    else:
        base = G.bign_big[str(l)]["base"]
        hs, ss, ps = (b"".join(bytes.fromhex(t[f]) for t in base) for f in ("hash", "sig", "pubkey"))
        m = len(base)
    for e in (8, 10, 11, 12, 13, 14):
        n = 1 << e
        reps = n // m + 1
        h, s, k = (torch.frombuffer(bytearray((x * reps)[: w * n]), dtype=torch.uint8).cuda()
                   for x, w in ((hs, l // 4), (ss, 3 * l // 8), (ps, l // 2)))
        codes = torch.empty(n, dtype=torch.int32, device="cuda")
        row = []
        for path in (0x43, 0x93, 0xA3):
            tune(2, path)
            ms = timed(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], h, s, k, codes))
            assert int((codes != 0).sum()) == 0, (l, e, hex(path), int((codes != 0).sum()))
            row.append(ms)
        tune(2, 0)
        print(f"l = {l}, 2^{e} signatures: quads {row[0]:.3f} ms, quad + helper {row[1]:.3f} ms (64 lanes per block, x{row[0] / row[1]:.2f}), "
              f"{row[2]:.3f} ms (256 lanes, x{row[0] / row[2]:.2f})", flush=True)
