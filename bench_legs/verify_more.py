"""bench leg: around verification -- bignPubkeyVal over 2^24 keys (the step `sig vfy` runs first), many signatures of ONE signer and
of 64 signers (key-table entries; no bee2 API, no BASELINE config: kept, frozen -- DESIGN.md 4.11).  Not in the default run."""
import ctypes
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, K, N = c.dist, c.eng, c.K, c.N
    others, hc, do_cpu = c.others, c.hc, c.do_cpu
    import goldenlib
    G = goldenlib.Golden()
    hs, ss, ps = G.bign_base_arrays()
    nbase = len(hs) // 32
    n = 1 << 18
    rng = np.random.default_rng(0xB164 + dist.rank)
    dk = torch.from_numpy(np.tile(np.frombuffer(ps, dtype=np.uint8), n // nbase).copy()).cuda()
    kv = max(3, min(K, 10))
    ms_general = others.get("bignVerify", {}).get("ms_per_step")     # the general entry over 2^18 signatures, when that leg ran
    # the step `sig vfy` runs before each verification: bign128PubkeyVal over the same keys (tiled 64x more: 1 GiB, beyond the 256 MiB MALL)
    kk = dk.repeat(64)
    nk = kk.numel() // 64
    kcodes = torch.empty(nk, dtype=torch.int32, device="cuda")
    el = timed(dist, kv, 2, lambda: eng.bignPubkeyValL_batch_dev(128, kk, kcodes))
    ach = 68 * nk / (timed.event_ms * 1e-3) / 1e9
    pv_traffic, pv_traffic_src = None, None
    others["bignPubkeyVal"] = {
        "metric": "bign-curve256v1 public keys validated/s", "value": N * nk * kv / el, "unit": "keys/s", "steps": kv,
        "ms_per_step": el / kv * 1e3, "all_valid": bool((kcodes == 0).all()),
        "config": {"workload": f"bignPubkeyVal batch: {nk} keys per GPU (the 2048 genuine keys tiled)"},
        "roofline": {"kernel": "bign_pubkey_val_kernel<8>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pv_traffic, "traffic_source": pv_traffic_src,
                     "avg_launch_ms": timed.event_ms,
                     "note": "68 B per key (64 in + 4 out); 2 squarings + 1 multiplication per key are ~0.3 ms of VALU work for "
                             "2^24 keys, i.e. arithmetic and HBM demands are about equal; wall time per launch"}}
    if do_cpu:
        import refgen
        if refgen.have_ref():
            ref = ctypes.CDLL(refgen.REF_SO)
            ref.bign128PubkeyVal.restype = ctypes.c_uint32
            t0, cnt = time.perf_counter(), 0
            while time.perf_counter() - t0 < 1.0:
                i = cnt % nbase
                ref.bign128PubkeyVal(ps[64 * i: 64 * i + 64])
                cnt += 1
            others["bignPubkeyVal"]["cpu_baseline"] = {
                "value": cnt / (time.perf_counter() - t0), "unit": "keys/s", "cores": 1, "kind": "reference",
                "sample": "1 s of bign128PubkeyVal calls, one thread"}
    # many signatures of ONE signer (the `sig vfy` batch over a tree of files): the key is a fixed base with a comb
    # table of its own, no doublings left.  2^18 DISTINCT valid signatures made here by the signing entry (untimed), 1/16 damaged.
    from bee2_amd.engine import OID_BELT_HASH_DER as _OID
    n1 = 1 << 18
    g1 = torch.Generator(device="cuda"); g1.manual_seed(0x51D + dist.rank)
    h1 = torch.empty(32 * n1, dtype=torch.uint8, device="cuda"); h1.view(torch.int64).random_(generator=g1)
    d1 = bytes(range(7, 39))[:31] + b"\x21"
    p1 = torch.empty(64, dtype=torch.uint8, device="cuda"); c1 = torch.empty(1, dtype=torch.int32, device="cuda")
    eng.bignPubkeyCalcL_batch_dev(128, torch.from_numpy(np.frombuffer(d1, dtype=np.uint8).copy()).cuda(), p1, c1)
    s1 = torch.empty(48 * n1, dtype=torch.uint8, device="cuda"); cs = torch.empty(n1, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(128, _OID, h1, torch.from_numpy(np.frombuffer(d1 * n1, dtype=np.uint8).copy()).cuda(), s1, cs)
    torch.cuda.synchronize()
    pub1 = p1.cpu().numpy().tobytes()
    bad1 = torch.from_numpy(rng.choice(n1, n1 // 16, replace=False)).cuda()
    s1.view(n1, 48)[bad1, 5] ^= 0x10
    codes1 = torch.empty(n1, dtype=torch.int32, device="cuda")
    # untimed: the first call builds and caches the key's 8-bit comb table; a key gets its 16-bit table once 2^19 signatures have
    # been verified under it -- two calls here -- so the timed calls see the steady state of a busy key (16 + 8 additions)
    for _ in range(3):
        eng.bignVerifyL_onekey_batch_dev(128, _OID, h1, s1, pub1, codes1)
    el = timed(dist, kv, 2, lambda: eng.bignVerifyL_onekey_batch_dev(128, _OID, h1, s1, pub1, codes1))
    ms1 = timed.event_ms
    want1 = torch.zeros(n1, dtype=torch.int32, device="cuda"); want1[bad1] = 510
    MADS_ONEKEY = 24 * 732 + 5 * 72          # 16 (u G) + 8 (v Q) mixed additions (8M + 3S) + x_R; inversions are division steps
    others["bignVerify_onekey"] = {
        "metric": "bign-curve256v1 verifies/s, one signer", "value": N * n1 * kv / el, "unit": "verifies/s", "steps": kv,
        "ms_per_step": el / kv * 1e3, "verdicts_as_expected": bool((codes1 == want1).all() and int(cs.abs().sum()) == 0),
        "vs_general_entry": (n1 * kv / el) / (n / (ms_general * 1e-3)) if ms_general else None,
        "config": {"workload": "bee2hip_bignVerifyL_onekey_batch_dev: 2^18 distinct signatures under ONE public key per GPU "
                               "(made by the signing entry, 1/16 damaged); the key's comb tables cached (16-bit windows: the state of a key after 2^19 signatures)"},
        "roofline": {"kernels": "bign_onekey + slow + inv + tail", "bound": "valu-int", "avg_batch_ms": ms1,
                     "mads_per_verify": MADS_ONEKEY, "achieved": MADS_ONEKEY * n1 / (ms1 * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                     "unit": "T v_mad_u64_u32 lane-ops/s", "frac": MADS_ONEKEY * n1 / (ms1 * 1e-3) / 1e12 / MAD_PEAK_T}}
    if dist.rank == 0:
        # the latency floor: prefixes of the same batch (up to 2^16 signatures four lanes share a signature)
        small1 = {}
        for e in (10, 14, 15, 16, 17):
            m = 1 << e
            pre = (h1[: 32 * m], s1[: 48 * m], pub1, codes1[:m])
            for _ in range(4):
                eng.bignVerifyL_onekey_batch_dev(128, _OID, *pre)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                eng.bignVerifyL_onekey_batch_dev(128, _OID, *pre)
            e1.record()
            torch.cuda.synchronize()
            ms_b = e0.elapsed_time(e1) / 20
            small1[f"2^{e}"] = {"ms_per_batch": ms_b, "verifies_per_s": m / (ms_b * 1e-3)}
        others["bignVerify_onekey"]["batch_size_sweep"] = small1
    # ... and of a FEW signers: the population SURVEY 8d describes (64 key pairs), every signature distinct, 1/16 damaged
    nk = 64
    dks = [bytes(((k * 37 + i * 11 + 5) & 255) for i in range(31)) + b"\x21" for k in range(nk)]
    pk = torch.empty(64 * nk, dtype=torch.uint8, device="cuda"); ck = torch.empty(nk, dtype=torch.int32, device="cuda")
    eng.bignPubkeyCalcL_batch_dev(128, torch.from_numpy(np.frombuffer(b"".join(dks), dtype=np.uint8).copy()).cuda(), pk, ck)
    kidx = torch.from_numpy(rng.integers(0, nk, n1).astype(np.int32)).cuda()
    dall = torch.from_numpy(np.frombuffer(b"".join(dks), dtype=np.uint8).copy()).cuda().view(nk, 32)[kidx.long()].reshape(-1).contiguous()
    eng.bignSign2L_batch_dev(128, _OID, h1, dall, s1, cs)
    torch.cuda.synchronize()
    s1.view(n1, 48)[bad1, 5] ^= 0x10
    pubs_k = pk.cpu().numpy().tobytes()
    eng.bignVerifyL_keyed_batch_dev(128, _OID, h1, s1, pubs_k, kidx, codes1)       # untimed: the 64 tables are built and cached
    el = timed(dist, kv, 2, lambda: eng.bignVerifyL_keyed_batch_dev(128, _OID, h1, s1, pubs_k, kidx, codes1))
    msk = timed.event_ms
    MADS_KEYED = 32 * 732 + 5 * 72           # 16 (u G) + 16 (v Q, 8-bit windows) mixed additions + x_R
    others["bignVerify_keyed"] = {
        "metric": "bign-curve256v1 verifies/s, 64 signers", "value": N * n1 * kv / el, "unit": "verifies/s", "steps": kv,
        "ms_per_step": el / kv * 1e3, "verdicts_as_expected": bool((codes1 == want1).all() and int(cs.abs().sum()) == 0 and int(ck.abs().sum()) == 0),
        "vs_general_entry": (n1 * kv / el) / (n / (ms_general * 1e-3)) if ms_general else None,
        "config": {"workload": "bee2hip_bignVerifyL_keyed_batch_dev: 2^18 distinct signatures of 64 signers per GPU (random signer per "
                               "signature, 1/16 damaged); the signers' 8-bit comb tables cached (17 MiB)"},
        "roofline": {"kernels": "bign_onekey<keyed> + slow + inv + tail", "bound": "valu-int", "avg_batch_ms": msk,
                     "mads_per_verify": MADS_KEYED, "achieved": MADS_KEYED * n1 / (msk * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                     "unit": "T v_mad_u64_u32 lane-ops/s", "frac": MADS_KEYED * n1 / (msk * 1e-3) / 1e12 / MAD_PEAK_T}}
    del h1, s1, cs, codes1, want1, kidx, dall, pk
    del dk, kk, kcodes

