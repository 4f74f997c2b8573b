import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, bee2_amd, goldenlib
from bee2_amd import engine as E
eng = bee2_amd.load(); eng.set_device(0)
G = goldenlib.Golden()
for l in (128, 192, 256):
    if l == 128:
        hs, ss, ps = G.bign_base_arrays(); hs, ss, ps = hs[:32 * 64], ss[:48 * 64], ps[:64 * 64]
    else:
        b = G.bign_big[str(l)]["base"][:64]
        hs, ss, ps = (b"".join(bytes.fromhex(t[k]) for t in b) for k in ("hash", "sig", "pubkey"))
    dev = lambda x: torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda()
    th, ts, tp = dev(hs), dev(ss), dev(ps)
    codes = torch.empty(64, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.bignVerifyL_batch_dev(l, E.LEVEL_OID[l], th, ts, tp, codes); torch.cuda.synchronize()
    t1 = time.perf_counter()
    eng.bignVerifyL_batch_dev(l, E.LEVEL_OID[l], th, ts, tp, codes); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"l={l}: first call {1e3*(t1-t0):.1f} ms (builds the comb table), second {1e3*(t2-t1):.2f} ms, all valid: {bool((codes == 0).all())}")
