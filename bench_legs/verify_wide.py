"""bench leg: the 384- and 512-bit curves"""
import ctypes
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, args, K, N, others, cores, do_cpu = c.dist, c.eng, c.args, c.K, c.N, c.others, c.cores, c.do_cpu
    from bee2_amd.engine import LEVEL_OID
    import goldenlib
    G = goldenlib.Golden()
    if "bign_big" not in G.__dict__:
        return
    for l in (192, 256):
        base = G.bign_big[str(l)]["base"]
        reps_l = (1 << 18) // len(base)
        hs_l = b"".join(bytes.fromhex(t["hash"]) for t in base) * reps_l
        ss_l = b"".join(bytes.fromhex(t["sig"]) for t in base) * reps_l
        ps_l = b"".join(bytes.fromhex(t["pubkey"]) for t in base) * reps_l
        nl = len(base) * reps_l
        th, ts, tp = (torch.from_numpy(np.frombuffer(x, dtype=np.uint8).copy()).cuda() for x in (hs_l, ss_l, ps_l))
        tc = torch.empty(nl, dtype=torch.int32, device="cuda")
        kl = max(2, min(K, 5))
        el = timed(dist, kl, 1, lambda: eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], th, ts, tp, tc))
        others[f"bignVerify_l{l}"] = {
            "metric": f"bign-curve{2 * l}v1 verifies/s", "value": N * nl * kl / el, "unit": "verifies/s",
            "steps": kl, "ms_per_step": el / kl * 1e3, "all_valid": bool((tc == 0).all()),
            "config": {"workload": f"{nl} signatures per GPU on the {2 * l}-bit curve (SURVEY 8f-4; 2^18 as configs[3]: 2^16 leaves one wavefront per SIMD), "
                                   f"{len(base)} genuine triples tiled"}}
        if dist.rank == 0:
            # small batches (prefixes of the same tiling): the quad kernel up to 2^14 signatures, the r01 kernels above
            small = {}
            no_l = l // 4
            for e in (10, 13, 14, 15):
                m = 1 << e
                args = (th[: no_l * m], ts[: (no_l + no_l // 2) * m], tp[: 2 * no_l * m], tc[:m])
                for _ in range(2):
                    eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], *args)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], *args)
                e1.record()
                torch.cuda.synchronize()
                ms_b = e0.elapsed_time(e1) / 5
                small[f"2^{e}"] = {"ms_per_batch": ms_b, "verifies_per_s": m / (ms_b * 1e-3)}
            others[f"bignVerify_l{l}"]["batch_size_sweep"] = small
        if do_cpu:
            import refgen
            if refgen.have_ref():
                ref = ctypes.CDLL(refgen.REF_SO)
                f = getattr(ref, f"bign{l}Verify")
                f.restype = ctypes.c_uint32
                no = l // 4
                t0, cnt = time.perf_counter(), 0
                while time.perf_counter() - t0 < 1.5:
                    i = cnt % len(base)
                    f(hs_l[no * i: no * i + no], ss_l[(no + no // 2) * i: (no + no // 2) * (i + 1)],
                      ps_l[2 * no * i: 2 * no * (i + 1)])
                    cnt += 1
                others[f"bignVerify_l{l}"]["cpu_baseline"] = {
                    "value": cnt / (time.perf_counter() - t0), "unit": "verifies/s", "cores": 1,
                    "kind": "reference", "sample": "1.5 s of bign%dVerify calls, one thread" % l}
        del th, ts, tp, tc

