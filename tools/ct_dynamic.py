"""Dynamic side of the constant-time check: sign the SAME hashes with private keys of a given class and let
rocprofv3 count the executed instructions of every signing kernel.  Secret-independent control flow means the
counts are identical for every class.  usage (under rocprofv3 --pmc ...): python tools/ct_dynamic.py <class> [l] [log2 n]
(n = 2^14 runs k G one lane per scalar, n <= 2^13 one wavefront per scalar: bign_mulbase_coop_kernel)
classes: random | small (d = 1 .. 16) | ones (d = q - 1 - i) | sparse (d = 2^k + 1) | dense (d = ~sparse mod q)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
import orclib  # noqa: E402
from bee2_amd import engine as E  # noqa: E402

cls = sys.argv[1]
l = int(sys.argv[2]) if len(sys.argv) > 2 else 128
no, sg = l // 4, 3 * l // 8
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 14)
eng = bee2_amd.load()
eng.set_device(0)
orc = orclib.load()
P = eng.bignParamsStd(E.CURVE_NAME[l])
q = int.from_bytes(bytes(P.q)[:no], "little")
if cls == "random":
    ds = [int.from_bytes(orc.fill(no, 1000 + i), "little") % (q - 1) + 1 for i in range(n)]
elif cls == "small":
    ds = [1 + i % 16 for i in range(n)]
elif cls == "ones":
    ds = [q - 1 - i for i in range(n)]
elif cls == "sparse":
    ds = [(1 << (i % (8 * no - 2))) + 1 for i in range(n)]
else:
    ds = [(q - 2 - ((1 << (i % (8 * no - 2))) + 1)) for i in range(n)]
privs = torch.frombuffer(bytearray(b"".join(d.to_bytes(no, "little") for d in ds)), dtype=torch.uint8).cuda()
hashes = torch.frombuffer(bytearray(orc.fill(no * n, 0xC7)), dtype=torch.uint8).cuda()
sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
pubs = torch.empty(2 * no * n, dtype=torch.uint8, device="cuda")
codes = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(2):
    eng.bignSign2L_batch_dev(l, E.LEVEL_OID[l], hashes, privs, sigs, codes)
    eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, codes)
torch.cuda.synchronize()
assert int(codes.abs().sum()) == 0

# round 3: the signing side on a NON-STANDARD parameter set (general-curve constant-time ladder, bign_generic_kernels.hip):
# the same key classes through the host batch entry points on an isomorphic image of the level's standard curve
import ctypes  # noqa: E402
import json  # noqa: E402
from bee2_amd.engine import bign_params  # noqa: E402
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "bign_generic.json")))
c = next(x for x in FIX["curves"] if x["kind"] == "iso" and x["l"] == l)
prm = bign_params()
prm.l = l
for f in ("p", "a", "b", "q", "yG"):
    raw = bytes.fromhex(c[f])
    ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
m = 512
gp = b"".join(d.to_bytes(no, "little") for d in ds[:m])
gh = orc.fill(no * m, 0xC7)
for _ in range(2):
    code, gs, gc = eng.bignSign2_batch(prm, E.LEVEL_OID[l], gh, gp, None)
    assert code == 0 and not any(gc)
    code, gpub, gcodes = eng.bignPubkeyCalc_batch(prm, gp)
    assert code == 0 and not any(gcodes)

# round 5: additional input LONGER than 64 octets -- theta = belt-hash(oid || d || t) for the whole batch by ONE ragged belt-hash launch
# on the bank-private S-box copies (belt_hash_ragged_kernel<BeltTabTwoP, 256>, launch_hash_ragged(..., secret = true)): the kernel's
# instruction counters and its LDS bank conflicts (0) must not depend on the key class either
Pstd = eng.bignParamsStd(E.CURVE_NAME[l])
tl = orc.fill(100, 0x7A)
for _ in range(2):
    code, ls_, lc = eng.bignSign2_batch(Pstd, E.LEVEL_OID[l], gh, gp, tl)
    assert code == 0 and not any(lc)
