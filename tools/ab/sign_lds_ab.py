#!/usr/bin/env python3
"""A/B of the one-lane k G forms of the signing side on the 256-bit curve (experiments build, bee2hip_internal_tune 10):
1 = signed 6-bit windows, masked scan of the row (round 3's product); 7 = signed 7-bit windows looked up in LDS, product
coordinates; 72 = the same with the other coordinate form.  Device-resident batches, hipEvents, outputs asserted identical.
usage: python tools/ab/sign_lds_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
from bee2_amd.engine import LEVEL_OID  # noqa: E402

eng = bee2_amd.load_experiments(); eng.set_device(0)
l, no, sg = 128, 32, 48


def ms(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


FORMS = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 0, 7, 8)      # (round 6: 81 = the 8-bit LDS form on 32-bit limbs, 8 / 0 = on 29-bit limbs)
print("ms per batch (pubkey calc | sign2) and M/s; forms: 1 = 6-bit scan, 0 = product dispatch by size (4 lanes up to 2^15, 8-bit LDS from 3*2^14: 512- / 1024-lane workgroups), 7 = 7-bit LDS, 8 = 8-bit LDS forced")
for e in (15, 16, 17, 18, 19):
    n = 1 << e
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    privs = torch.empty(no * n, dtype=torch.uint8, device="cuda"); privs.view(torch.int64).random_(generator=g)
    privs.view(-1, no)[:, no - 1] &= 0x7F
    privs.view(-1, no)[:, 0] |= 1
    hsh = torch.empty(no * n, dtype=torch.uint8, device="cuda"); hsh.view(torch.int64).random_(generator=g)
    out = {}
    line = f"2^{e}: "
    for form in FORMS:
        eng.lib.bee2hip_internal_tune(10, form)
        pubs = torch.zeros(2 * no * n, dtype=torch.uint8, device="cuda")
        sigs = torch.zeros(sg * n, dtype=torch.uint8, device="cuda")
        sc = torch.empty(n, dtype=torch.int32, device="cuda")
        t_pk = ms(lambda: eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, sc))
        t_sg = ms(lambda: eng.bignSign2L_batch_dev(l, LEVEL_OID[l], hsh, privs, sigs, sc))
        out[form] = (pubs.cpu().numpy().tobytes(), sigs.cpu().numpy().tobytes())
        line += f" form {form:2d}: {t_pk:.3f} | {t_sg:.3f} ms  ({n / t_pk / 1e3:6.1f} | {n / t_sg / 1e3:6.1f} M/s)"
    eng.lib.bee2hip_internal_tune(10, 0)
    assert all(out[f] == out[FORMS[0]] for f in FORMS), e
    print(line)
