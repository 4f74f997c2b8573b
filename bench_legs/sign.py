"""bench leg: key generation / signing, constant-time kernels (SURVEY 8f-4 tail)"""
import ctypes
import time

import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, K, N, others, cores, do_cpu = c.dist, c.eng, c.K, c.N, c.others, c.cores, c.do_cpu
    from bee2_amd.engine import LEVEL_OID
    l, no, sg = 128, 32, 48
    n = 1 << 18
    privs = torch.empty(no * n, dtype=torch.uint8, device="cuda")
    fill_seeded(privs, 0x5164 + dist.rank)
    privs.view(-1, no)[:, no - 1] &= 0x7F                       # d < 2^255 < q: every key valid
    privs.view(-1, no)[:, 0] |= 1
    hsh = torch.empty(no * n, dtype=torch.uint8, device="cuda")
    fill_seeded(hsh, 0x5165 + dist.rank)
    sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
    pubs = torch.empty(2 * no * n, dtype=torch.uint8, device="cuda")
    sc = torch.empty(n, dtype=torch.int32, device="cuda")
    ks = max(3, min(K, 10))
    el = timed(dist, ks, 2, lambda: eng.bignSign2L_batch_dev(l, LEVEL_OID[l], hsh, privs, sigs, sc))
    ms_sign = timed.event_ms
    el_k = timed(dist, ks, 2, lambda: eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, sc))
    ms_calc = timed.event_ms
    vc = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_batch_dev(l, LEVEL_OID[l], hsh, sigs, pubs, vc)        # not timed: every signature must verify
    torch.cuda.synchronize()
    # 32x32+64 multiply-adds per signature: 33 signed 8-bit windows (round 4; 43 of 6 bits in round 3, 64 of 4 bits before) x 11
    # multiplications x (64 + 8) -- see below for the count of the Jacobian form -- and the affine coordinates; the inversion (fixed-count division steps,
    # fe_inv_safegcd<N, true>), the belt work (16 block encryptions) and the table scan have none
    # (8 M + 3 S per Jacobian mixed addition: 8 x 72 + 3 x 52 multiply-adds with the reductions; 4 M + 1 S for x, y)
    mads = 33 * (8 * 72 + 3 * 52) + 2 * 72 + 52     # round 4: 33 signed 8-bit windows looked up in LDS (bign_mulbase_lds_kernel); x_R only
    others["bignSign2"] = {
        "metric": "bign-curve256v1 deterministic signatures/s", "value": N * n * ks / el, "unit": "signatures/s", "steps": ks,
        "ms_per_step": el / ks * 1e3, "all_verify": bool((vc == 0).all() and (sc == 0).all()),
        "config": {"workload": f"bignSign2 batch: {n} (hash, private key) pairs per GPU on bign-curve256v1, no additional input; "
                               "constant-time kernels (nonce by belt-hash + belt-wbl, k G on signed 8-bit windows whose entry is looked up in bank-private LDS copies of the row, "
                               "masked Jacobian mixed additions, inversion by a fixed number of division steps); every signature verified afterwards (untimed)"},
        "roofline": {"kernels": "bign_sign_nonce + bign_mulbase_lds (one lane per signature, 29-bit limbs, window entries looked up in LDS, 1/Z shared in the workgroup) + bign_sign_tail", "bound": "valu-int", "avg_batch_ms": ms_sign,
                     "mads_per_signature": mads, "achieved": mads * n / (ms_sign * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                     "unit": "T v_mad_u64_u32 lane-ops/s", "frac": mads * n / (ms_sign * 1e-3) / 1e12 / MAD_PEAK_T,
                     "note": "same multiplier formulation as verification (each mad paired with a half-rate addc: 0.5 is the ceiling)"},
        "pubkey_calc": {"value": N * n * ks / el_k, "unit": "keys/s", "ms_per_step": el_k / ks * 1e3, "avg_batch_ms": ms_calc},
    }
    if dist.rank == 0:
        # prefixes of the same device-resident batch: k G runs on 64 / 16 / 4 lanes per signature up to 2^10 / 2^13 / 2^16
        # signatures, one lane above (profiles/r03_sign_coop.txt); wall clock around launch + synchronise, best of 5
        small = {}
        for e in (0, 10, 12, 14, 15, 16):
            m = 1 << e
            best = 1e9
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.bignSign2L_batch_dev(l, LEVEL_OID[l], hsh[: no * m], privs[: no * m], sigs[: sg * m], sc[:m])
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            small[f"2^{e}"] = {"ms_per_batch": best * 1e3, "signatures_per_s": m / best}
        others["bignSign2"]["batch_size_sweep"] = small
    if do_cpu:
        import refgen
        if refgen.have_ref():
            ref = ctypes.CDLL(refgen.REF_SO)
            ref.bign128Sign2.restype = ctypes.c_uint32
            hh, pp = hsh[: no * 512].cpu().numpy().tobytes(), privs[: no * 512].cpu().numpy().tobytes()
            out = ctypes.create_string_buffer(sg)
            t0, cnt = time.perf_counter(), 0
            while time.perf_counter() - t0 < 3.0:
                i = cnt % 512
                ref.bign128Sign2(out, hh[no * i: no * i + no], pp[no * i: no * i + no], None, ctypes.c_size_t(0))
                cnt += 1
            others["bignSign2"]["cpu_baseline"] = {
                "value": cnt / (time.perf_counter() - t0), "unit": "signatures/s", "cores": 1, "kind": "reference",
                "sample": "3 s of bign128Sign2 calls on 512 of the same (hash, key) pairs, one thread"}
    del privs, hsh, sigs, pubs, sc, vc

