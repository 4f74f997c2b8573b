#!/usr/bin/env python3
"""bee2hip_bignVerifyL_keyed_batch_dev: wall time of the FIRST call with K keys the library has not met (their tables are built together:
one allocation, one upload, one kernel launch) and of the same call again, against the general entry.  usage: python tools/ab/keyed_first_call.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402
from bee2_amd.engine import LEVEL_OID  # noqa: E402

eng = bee2_amd.load(); eng.set_device(0)
l, no, sg = 128, 32, 48
oid = LEVEL_OID[l]
rng = np.random.default_rng(7)
# warm: G's tables, the kernels' code, the scratch
w = torch.zeros(no * 256, dtype=torch.uint8, device="cuda"); ws = torch.zeros(sg * 256, dtype=torch.uint8, device="cuda")
wc = torch.empty(256, dtype=torch.int32, device="cuda")
eng.bignVerifyL_onekey_batch_dev(l, oid, w, ws, bytes(31) + b"\x00" + bytes(32), wc)   # (a key off the curve: the general path, G's table)
for nk, n in ((1, 1 << 12), (8, 1 << 14), (64, 1 << 16), (64, 1 << 18), (1024, 1 << 18)):
    dks = [bytes(rng.integers(0, 256, no - 1, dtype=np.uint8)) + b"\x21" for _ in range(nk)]
    pk = torch.empty(2 * no * nk, dtype=torch.uint8, device="cuda"); ck = torch.empty(nk, dtype=torch.int32, device="cuda")
    dk = torch.from_numpy(np.frombuffer(b"".join(dks), dtype=np.uint8).copy()).cuda()
    eng.bignPubkeyCalcL_batch_dev(l, dk, pk, ck)
    kidx = torch.from_numpy(rng.integers(0, nk, n).astype(np.int32)).cuda()
    h = torch.from_numpy(rng.integers(0, 256, no * n, dtype=np.uint8)).cuda()
    s = torch.empty(sg * n, dtype=torch.uint8, device="cuda"); cs = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(l, oid, h, dk.view(nk, no)[kidx.long()].reshape(-1).contiguous(), s, cs)
    torch.cuda.synchronize()
    pubs = pk.cpu().numpy().tobytes()
    allk = pk.view(nk, 2 * no)[kidx.long()].reshape(-1).contiguous()
    codes = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_batch_dev(l, oid, h, s, allk, codes); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.bignVerifyL_batch_dev(l, oid, h, s, allk, codes); torch.cuda.synchronize(); tg = (time.perf_counter() - t0) * 1e3
    assert not codes.any()
    t0 = time.perf_counter(); eng.bignVerifyL_keyed_batch_dev(l, oid, h, s, pubs, kidx, codes); torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) * 1e3
    assert not codes.any()
    t0 = time.perf_counter(); eng.bignVerifyL_keyed_batch_dev(l, oid, h, s, pubs, kidx, codes); torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) * 1e3
    assert not codes.any()
    print(f"{nk:5d} signers, {n:7d} signatures: first keyed call {t1:7.3f} ms wall (tables built), again {t2:6.3f} ms, general entry {tg:6.3f} ms")
