#!/usr/bin/env python3
"""n signatures under ONE key (bee2hip_bignVerifyL_onekey_batch_dev) against the general batch entry with the key repeated, by batch
size and curve: ms per device-resident batch (hipEvents), the first call with a new key (table construction included) apart.
Distinct valid signatures made by the signing entry.  usage: [ONEKEY_TAB16=63|0] python tools/ab/onekey_sizes.py [max_log2 = 20]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
from bee2_amd.engine import LEVEL_OID  # noqa: E402

emax = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if os.environ.get("ONEKEY_TAB16"):       # experiments build: 63 = the 8-bit table of a key only, 0 = the 16-bit one from the first signature
    eng = bee2_amd.load_experiments(); eng.set_device(0)
    eng.lib.bee2hip_internal_tune(20, int(os.environ["ONEKEY_TAB16"]))
    print(f"# experiments build, 16-bit table of a key after 2^{os.environ['ONEKEY_TAB16']} signatures")
else:
    eng = bee2_amd.load(); eng.set_device(0)


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


def signed(l, n, seed):
    no, sg = l // 4, 3 * l // 8
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    h = torch.empty(no * n, dtype=torch.uint8, device="cuda"); h.view(torch.int64).random_(generator=g)
    d = bytes((seed * 7 + i) & 255 for i in range(no - 1)) + b"\x21"
    p = torch.empty(2 * no, dtype=torch.uint8, device="cuda"); c1 = torch.empty(1, dtype=torch.int32, device="cuda")
    eng.bignPubkeyCalcL_batch_dev(l, torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).cuda(), p, c1)
    s = torch.empty(sg * n, dtype=torch.uint8, device="cuda"); cs = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(l, LEVEL_OID[l], h, torch.from_numpy(np.frombuffer(d * n, dtype=np.uint8).copy()).cuda(), s, cs)
    torch.cuda.synchronize()
    assert int(cs.abs().sum()) == 0 and int(c1.item()) == 0
    return h, s, p.cpu().numpy().tobytes()


for l in (128, 192, 256):
    no, sg = l // 4, 3 * l // 8
    nmax = 1 << (emax if l == 128 else min(emax, 18))
    h, s, pub = signed(l, nmax, 11 + l)
    keys = torch.from_numpy(np.frombuffer(pub, dtype=np.uint8).copy()).cuda().repeat(nmax)
    codes = torch.empty(nmax, dtype=torch.int32, device="cuda")
    # the first call with a key the library has not seen: host base points + table kernel + the batch
    h2, s2, pub2 = signed(l, 1024, 99 + l)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.bignVerifyL_onekey_batch_dev(l, LEVEL_OID[l], h2, s2, pub2, codes[:1024]); torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    eng.bignVerifyL_onekey_batch_dev(l, LEVEL_OID[l], h2, s2, pub2, codes[:1024]); torch.cuda.synchronize()
    again = (time.perf_counter() - t0) * 1e3
    assert not codes[:1024].any()
    print(f"l = {l}: 1024 signatures under a NEW key {first:.3f} ms wall, the same call again {again:.3f} ms")
    for e in range(10, 21):
        n = 1 << e
        if n > nmax:
            break
        a = (h[: no * n], s[: sg * n])
        ms1 = t(lambda: eng.bignVerifyL_onekey_batch_dev(l, LEVEL_OID[l], a[0], a[1], pub, codes[:n]))
        assert not codes[:n].any()
        msg = t(lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], a[0], a[1], keys[: 2 * no * n], codes[:n]), reps=5)
        assert not codes[:n].any()
        print(f"  2^{e}: one key {ms1:.3f} ms ({n / ms1 / 1e3:7.1f} M/s)   general {msg:.3f} ms ({n / msg / 1e3:6.1f} M/s)   x{msg / ms1:.2f}")
