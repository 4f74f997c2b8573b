"""-m gpu: GF(2^256-189) layer and batched bign verify (mirrors
test/crypto/bign_test.c:338-357,388-400, bign128_test.c:101-163, math/zz_test.c)."""
import ctypes
import random

import numpy as np
import pytest
import torch

from bee2_amd import engine as E
from gpulib import dev, engine, exp_engine, host

pytestmark = pytest.mark.gpu
P = 2 ** 256 - 189


def _fe_run(eng, op, A, B):
    ta = dev(b"".join(x.to_bytes(32, "little") for x in A))
    tb = dev(b"".join(x.to_bytes(32, "little") for x in B))
    out = torch.empty_like(ta)
    code = eng.lib.bee2hip_debug_fe(op, ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()),
                                    ctypes.c_void_p(out.data_ptr()), ctypes.c_size_t(len(A)), None)
    assert code == 0
    torch.cuda.synchronize()
    raw = host(out)
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def test_field_ops_vs_python_ints():
    """regular GF(p) routines against big-int arithmetic on random and boundary values,
    including non-canonical inputs in [p, 2^256) (the kernels keep values weakly reduced)"""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    rnd = random.Random(1)
    special = [0, 1, 2, 188, 189, 190, P - 2, P - 1, P, P + 1, P + 188, 2 ** 256 - 1, 2 ** 255,
               2 ** 128, 2 ** 128 - 1, 2 ** 224 - 1, (1 << 256) - (1 << 32)]
    pool = special + [rnd.getrandbits(256) for _ in range(300)]
    A = special * len(special) + [rnd.choice(pool) for _ in range(4096)]
    B = [b for b in special for _ in special] + [rnd.choice(pool) for _ in range(4096)]
    ops = {0: lambda a, b: a * b % P, 1: lambda a, b: a * a % P, 2: lambda a, b: (a + b) % P,
           3: lambda a, b: (a - b) % P, 4: lambda a, b: pow(a, P - 2, P), 5: lambda a, b: 3 * a * b % P,
           6: lambda a, b: 8 * a * a % P, 7: lambda a, b: a % P,
           # 11: inversion by division steps alone, 12: the checked inversion the kernels call
           11: lambda a, b: pow(a, P - 2, P), 12: lambda a, b: pow(a, P - 2, P)}
    A += [2 ** k for k in range(256)] + [P - 2 ** k for k in range(255)] + [2 ** k - 1 for k in range(1, 256)]
    B += [0] * (len(A) - len(B))
    for op, f in ops.items():
        got = _fe_run(eng, op, A, B)
        want = [f(a, b) for a, b in zip(A, B)]
        assert got == want, f"field op {op}"


LAZY = {128: (9, 29, 189), 192: (14, 28, 317), 256: (19, 27, 569)}      # limbs, bits per limb, c of p = 2^(2l) - c


def _fe_run_l(eng, l, op, A, B):
    nb = l // 4
    ta = dev(b"".join(x.to_bytes(nb, "little") for x in A))
    tb = dev(b"".join(x.to_bytes(nb, "little") for x in B))
    out = torch.empty_like(ta)
    code = eng.lib.bee2hip_debug_feL(ctypes.c_size_t(l), op, ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()),
                                     ctypes.c_void_p(out.data_ptr()), ctypes.c_size_t(len(A)), None)
    assert code == 0
    torch.cuda.synchronize()
    raw = host(out)
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


def _limb_patterns(rnd, l):
    """2l-bit values whose lazy limbs sit at the corners the carry-free accumulation has to survive"""
    L, Bb, _ = LAZY[l]
    M = (1 << Bb) - 1
    mask = (1 << (2 * l)) - 1
    vals = []
    for pat in ([M] * L, [0] * L, [M, 0] * L, [0, M] * L, [1] * L, [M - 1] * L, [M] + [0] * (L - 1), [0] * (L - 1) + [M]):
        vals.append(sum(x << (Bb * i) for i, x in enumerate(pat[:L])) & mask)
    for _ in range(64):
        limbs = [rnd.choice([0, 1, M, M - 1, rnd.getrandbits(Bb)]) for _ in range(L)]
        vals.append(sum(x << (Bb * i) for i, x in enumerate(limbs)) & mask)
    return vals


@pytest.mark.parametrize("l", [128, 192, 256])
def test_field_ops_lazy_limbs_vs_python_ints(l):
    """bign_fe29.hpp (the small-batch form: signed 29 / 28 / 27-bit limbs on the three curves, lazy additions) against
    big-int arithmetic: multiplication / squaring with the folded small multiples, lazy differences and sums fed
    straight into multiplications at the bounds the point formulas use, the exact conversion back to 32-bit words"""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    rnd = random.Random(29 + l)
    L, Bb, c = LAZY[l]
    P = 2 ** (2 * l) - c
    special = [0, 1, 2, c - 1, c, c + 1, P - 2, P - 1, P, P + 1, P + c - 1, 2 ** (2 * l) - 1, 2 ** (2 * l - 1),
               2 ** (Bb * (L - 1)), 2 ** (Bb * (L - 1)) - 1, 2 ** Bb - 1, 2 ** Bb, (1 << (2 * l)) - (1 << 32)] + _limb_patterns(rnd, l)
    pool = special + [rnd.getrandbits(2 * l) for _ in range(300)]
    extra = 8192 if l == 128 else 2048
    A = [a for a in special for _ in special] + [rnd.choice(pool) for _ in range(extra)]
    B = [b for _ in special for b in special] + [rnd.choice(pool) for _ in range(extra)]
    ops = {20: lambda a, b: a * b % P, 21: lambda a, b: a * a % P, 22: lambda a, b: 3 * a * b % P,
           23: lambda a, b: 8 * a * a % P, 24: lambda a, b: (a - b) % P, 25: lambda a, b: 3 * (a - b) * (a + b) % P,
           26: lambda a, b: 4 * (a - 3 * b) * (-b) % P, 27: lambda a, b: 8 * (a - b) ** 2 % P,
           28: lambda a, b: (2 * (a - b) * (a + b) - 3 * a) % P}
    for op, f in ops.items():
        got = _fe_run_l(eng, l, op, A, B)
        want = [f(a, b) for a, b in zip(A, B)]
        bad = [(hex(a), hex(b)) for a, b, g, w in zip(A, B, got, want) if g != w]
        assert not bad, f"l = {l}, lazy-limb field op {op}: {len(bad)} wrong, first {bad[0]}"


@pytest.mark.parametrize("l", [128, 192, 256])
def test_point_ops_lazy_limbs_match_the_group_law(golden, l):
    """2P and 3P = 2P + P of public keys from the fixtures: affine x from jac29_dbl / jac29_madd (one lane per point)
    and from quad29_dbl / quad29_add (a DPP quad per point: every point is passed four times) == the Python group law"""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    c = LAZY[l][2]
    P = 2 ** (2 * l) - c
    no = l // 4
    if l == 128:
        _, _, ps = golden.bign_base_arrays()
        keys = [ps[64 * i:64 * i + 64] for i in range(256)]
    else:
        keys = [bytes.fromhex(t["pubkey"]) for t in golden.bign_big[str(l)]["base"]]
    pts = [(int.from_bytes(k[:no], "little"), int.from_bytes(k[no:], "little")) for k in keys]

    def dbl(x, y):
        lam = (3 * x * x - 3) * pow(2 * y, P - 2, P) % P
        x3 = (lam * lam - 2 * x) % P
        return x3, (lam * (x - x3) - y) % P

    def add(x1, y1, x2, y2):
        lam = (y2 - y1) * pow(x2 - x1, P - 2, P) % P
        x3 = (lam * lam - x1 - x2) % P
        return x3, (lam * (x1 - x3) - y1) % P

    X = [p[0] for p in pts]
    Y = [p[1] for p in pts]
    want2 = [dbl(x, y)[0] for x, y in pts]
    want3 = [add(*dbl(x, y), x, y)[0] for x, y in pts]
    assert _fe_run_l(eng, l, 29, X, Y) == want2
    assert _fe_run_l(eng, l, 30, X, Y) == want3
    X4 = [x for x in X for _ in range(4)]
    Y4 = [y for y in Y for _ in range(4)]
    assert _fe_run_l(eng, l, 31, X4, Y4) == [w for w in want2 for _ in range(4)]
    assert _fe_run_l(eng, l, 32, X4, Y4) == [w for w in want3 for _ in range(4)]


@pytest.mark.parametrize("path", [1, 2, 3, 0x43, 0x23, 0x83])
def test_bign_both_main_kernels_on_edge_and_base_sets(golden, path):
    """The batch size picks the main kernel (29-bit limbs up to 2^16 signatures, 32-bit above); here each is FORCED
    (bee2hip_internal_tune(2, path): 1 = 32-bit limbs, 2 = 29-bit limbs, 3 = one signature per quad or pair of lanes
    by size, 0x43 = quads, 0x23 = pairs, 0x83 = quad + helper quad) over the 433 edge cases (exceptional group-law cases
    included: they must reach the slow path from either kernel), the valid base set, and a 70 000-signature tiling
    with every 7th signature corrupted -- a size the 29-bit kernel never sees unforced."""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    tune = eng.lib.bee2hip_internal_tune
    tune.restype = ctypes.c_uint32
    assert tune(2, path) == 0
    try:
        Ecases = golden.bign_edge
        eh = b"".join(bytes.fromhex(e["hash"]) for e in Ecases)
        es = b"".join(bytes.fromhex(e["sig"]) for e in Ecases)
        ep = b"".join(bytes.fromhex(e["pubkey"]) for e in Ecases)
        codes = torch.full((len(Ecases),), -1, dtype=torch.int32, device="cuda")
        eng.bign128Verify_batch_dev(dev(eh), dev(es), dev(ep), codes)
        torch.cuda.synchronize()
        got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
        bad = [(e["name"], g, e["code"]) for e, g in zip(Ecases, got) if g != e["code"]]
        assert not bad, bad[:10]
        hs, ss, ps = golden.bign_base_arrays()
        n0 = len(hs) // 32
        reps = 70_000 // n0 + 1
        H = np.frombuffer(hs * reps, dtype=np.uint8).copy()
        S = np.frombuffer(ss * reps, dtype=np.uint8).copy()
        K = np.frombuffer(ps * reps, dtype=np.uint8).copy()
        n = 70_000
        H, S, K = H[: 32 * n], S[: 48 * n], K[: 64 * n]
        S.reshape(n, 48)[::7, 5] ^= 0x40                      # s0 of every 7th signature
        codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        eng.bign128Verify_batch_dev(torch.from_numpy(H).cuda(), torch.from_numpy(S).cuda(), torch.from_numpy(K).cuda(), codes)
        torch.cuda.synchronize()
        got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        want = np.zeros(n, dtype=np.int64)
        want[::7] = 510
        assert np.array_equal(got, want)
    finally:
        tune(2, 0)


def test_bign_G2_G3_dropin(golden):
    eng = engine()
    params = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    for k in golden.kat["bign_verify"]:
        h, s, p = (bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey"))
        assert eng.bign128Verify(h, s, p) == k["code"], k["name"]
        assert eng.bignVerify(params, E.OID_BELT_HASH_DER, h, s, p) == k["code"], k["name"]


def test_bign_base_set_all_valid(golden):
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    n = len(hs) // 32
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(hs), dev(ss), dev(ps), codes)
    torch.cuda.synchronize()
    assert int((codes != 0).sum()) == 0
    code, host_codes = eng.bignVerify_batch(hs[: 32 * 100], ss[: 48 * 100], ps[: 64 * 100])
    assert code == 0 and host_codes == [0] * 100


def test_bign_edge_cases_match_reference_codes(golden):
    """Q = +-G and small multiples, R = O, s1 >= q, coordinates >= p, off-curve Q, H >= q,
    bit flips: the err_t of the reference for each (tools/make_golden.py)"""
    eng = engine()
    Ecases = golden.bign_edge
    eh = b"".join(bytes.fromhex(e["hash"]) for e in Ecases)
    es = b"".join(bytes.fromhex(e["sig"]) for e in Ecases)
    ep = b"".join(bytes.fromhex(e["pubkey"]) for e in Ecases)
    codes = torch.full((len(Ecases),), -1, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(eh), dev(es), dev(ep), codes)
    torch.cuda.synchronize()
    got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    bad = [(e["name"], g, e["code"]) for e, g in zip(Ecases, got) if g != e["code"]]
    assert not bad, bad[:10]
    assert {0, 505, 510} <= set(got)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 257, 1000])
def test_bign_ragged_batch_sizes(orc, golden, n):
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    hs, ss, ps = bytearray(hs[: 32 * n]), bytearray(ss[: 48 * n]), bytearray(ps[: 64 * n])
    for i in range(0, n, 3):
        ss[48 * i + (i % 48)] ^= 0x10
    codes = torch.full((n + 8,), 0x7777, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(hs), dev(ss), dev(ps), codes)
    torch.cuda.synchronize()
    got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    assert got[:n] == orc.verify_batch(hs, ss, ps, nthreads=8)
    assert got[n:] == [0x7777] * 8


def test_bign_other_oid_and_errors(orc, golden):
    eng = engine()
    params = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    h, s, p = golden.bign_base[0]
    # a different (valid) OID changes the hashed message: the belt-hash OID signature fails
    other = bytes.fromhex("06092A7000020022651F52")
    assert eng.bignVerify(params, other, h, s, p) == E.ERR_BAD_SIG
    assert eng.bignVerify(params, b"\x06\x01", h, s, p) == E.ERR_BAD_OID
    code, _ = eng.bignVerify_batch(h, s, p, oid_der=b"\x07\x01\x00")
    assert code == E.ERR_BAD_OID


def test_bign_full_size_2pow18_tiled_and_corrupted(orc, golden):
    """BASELINE.json configs[3]: 2^18 signatures.  The 2048 genuine triples are tiled
    128x and a seeded 1/16 of the entries corrupted; every corrupted entry
    is checked against the oracle, every untouched one must verify."""
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    nb = len(hs) // 32
    reps = (1 << 18) // nb
    H = np.tile(np.frombuffer(hs, dtype=np.uint8), reps).reshape(-1, 32).copy()
    S = np.tile(np.frombuffer(ss, dtype=np.uint8), reps).reshape(-1, 48).copy()
    K = np.tile(np.frombuffer(ps, dtype=np.uint8), reps).reshape(-1, 64).copy()
    n = H.shape[0]
    rnd = random.Random(0xB164)
    bad_idx = sorted(rnd.sample(range(n), n // 16))
    for i in bad_idx:
        kind = rnd.randrange(4)
        if kind == 0:
            S[i, rnd.randrange(16)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            S[i, 16 + rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            H[i, rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        else:
            K[i, rnd.randrange(64)] ^= 1 << rnd.randrange(8)
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(H.reshape(-1)), dev(S.reshape(-1)), dev(K.reshape(-1)), codes)
    torch.cuda.synchronize()
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    want_bad = orc.verify_batch(H[bad_idx].tobytes(), S[bad_idx].tobytes(), K[bad_idx].tobytes(), nthreads=16)
    assert [int(got[i]) for i in bad_idx] == want_bad
    mask = np.ones(n, dtype=bool)
    mask[bad_idx] = False
    assert int((got[mask] != 0).sum()) == 0
    assert set(want_bad) <= {0, 505, 510} and 510 in want_bad


def test_bign_big_host_batch_is_uploaded_in_chunks(golden):
    """bee2hip_bignVerify_batch with host pointers and 2^19 + 777 signatures: chunks of 2^18 are uploaded while the previous
    chunk is verified (capi.hip, knob 11); the codes are those of the one-piece path and of the device-resident entry"""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    hs, ss, ps = golden.bign_base_arrays()
    nb = len(hs) // 32
    n = (1 << 19) + 777
    reps = (n + nb - 1) // nb
    H = np.tile(np.frombuffer(hs, dtype=np.uint8), reps).reshape(-1, 32)[:n].copy()
    S = np.tile(np.frombuffer(ss, dtype=np.uint8), reps).reshape(-1, 48)[:n].copy()
    K = np.tile(np.frombuffer(ps, dtype=np.uint8), reps).reshape(-1, 64)[:n].copy()
    rnd = random.Random(0x19)
    bad = sorted(rnd.sample(range(n), 5000) + [0, (1 << 18) - 1, 1 << 18, (1 << 19) - 1, 1 << 19, n - 1])
    for i in bad:
        S[i, rnd.randrange(48)] ^= 1 << rnd.randrange(8)
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(H.reshape(-1)), dev(S.reshape(-1)), dev(K.reshape(-1)), codes)
    torch.cuda.synchronize()
    want = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    assert all(want[i] != 0 for i in bad) and sum(1 for c in want if c) == len(set(bad))
    hb, sb, kb = H.tobytes(), S.tobytes(), K.tobytes()
    for knob in (1, 0):
        eng.lib.bee2hip_internal_tune(11, knob)
        code, got = eng.bignVerify_batch(hb, sb, kb)
        assert code == 0 and got == want, knob
    eng.lib.bee2hip_internal_tune(11, 1)


def test_concurrent_batches_on_two_streams_do_not_share_scratch(orc, golden):
    """two verify batches in flight on different streams (library scratch is per stream)"""
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    n = 2048
    bad = bytearray(ss)
    for i in range(n):
        bad[48 * i] ^= 1                              # every signature of the second batch is invalid
    dh, ds, dp, db = dev(hs), dev(ss), dev(ps), dev(bytes(bad))
    c1 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    c2 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            eng.bign128Verify_batch_dev(dh, ds, dp, c1)
        with torch.cuda.stream(s2):
            eng.bign128Verify_batch_dev(dh, db, dp, c2)
    torch.cuda.synchronize()
    assert int((c1 != 0).sum()) == 0
    assert int((c2 != 510).sum()) == 0


def test_oid_der_validation_matches_reference(golden):
    """valid / invalid verdicts of oidFromDER (tests/golden/oid_der_cases.json) through bignVerify:
    invalid -> ERR_BAD_OID, valid -> the signature check runs (here: ERR_BAD_SIG for a null signature,
    or ERR_NOT_IMPLEMENTED for an OID longer than the kernel stages)"""
    import json
    import os
    eng = engine()
    p = eng.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oid_der_cases.json")
    h, s, k = golden.bign_base[0]
    for c in json.load(open(path)):
        der = bytes.fromhex(c["der"])
        got = eng.bignVerify(p, der, h, s, k)
        if not c["valid"]:
            assert got == E.ERR_BAD_OID, c["der"]
        elif der == E.OID_BELT_HASH_DER:
            assert got == 0
        else:
            assert got in (E.ERR_BAD_SIG, E.ERR_NOT_IMPLEMENTED), (c["der"], got)


@pytest.mark.parametrize("l", [192, 256])
def test_field_ops_big_curves(l):
    """GF(2^384 - 317) and GF(2^512 - 569) against Python integers"""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    P = 2 ** (2 * l) - {192: 317, 256: 569}[l]
    nb = l // 4
    rnd = random.Random(l)
    special = [0, 1, 2, P - 1, P, P + 1, 2 ** (2 * l) - 1, 2 ** (2 * l - 1), 2 ** l, 316, 317, 569, 570]
    pool = special + [rnd.getrandbits(2 * l) for _ in range(100)]
    A = special * len(special) + [rnd.choice(pool) for _ in range(1024)]
    B = [b for b in special for _ in special] + [rnd.choice(pool) for _ in range(1024)]
    A += [2 ** k for k in range(2 * l)] + [P - 2 ** k for k in range(2 * l - 1)] + [2 ** k - 1 for k in range(1, 2 * l)]
    B += [0] * (len(A) - len(B))
    ta = dev(b"".join(x.to_bytes(nb, "little") for x in A))
    tb = dev(b"".join(x.to_bytes(nb, "little") for x in B))
    out = torch.empty_like(ta)
    ops = {0: lambda a, b: a * b % P, 1: lambda a, b: a * a % P, 2: lambda a, b: (a + b) % P,
           3: lambda a, b: (a - b) % P, 4: lambda a, b: pow(a, P - 2, P), 5: lambda a, b: 3 * a * b % P,
           6: lambda a, b: 8 * a * a % P, 7: lambda a, b: a % P,
           11: lambda a, b: pow(a, P - 2, P), 12: lambda a, b: pow(a, P - 2, P)}
    for op, f in ops.items():
        code = eng.lib.bee2hip_debug_feL(ctypes.c_size_t(l), op, ctypes.c_void_p(ta.data_ptr()),
                                         ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_size_t(len(A)), None)
        assert code == 0
        torch.cuda.synchronize()
        raw = host(out)
        got = [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]
        assert got == [f(a, b) for a, b in zip(A, B)], f"l={l} field op {op}"


@pytest.mark.parametrize("l", [128, 192, 256])
def test_crandall_reduction_carry_ripple(l):
    """fe_reduce on raw 2N-limb values (debug ops 9 / 10 = K 1 / 3) crafted so that the second fold
    lands within +-2 of 2^(32N): the carry of `t[0] + c*C` then ripples through every limb and (for
    delta >= 0) out of the top.  Random products almost never exercise that chain."""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    rnd = random.Random(l)
    C = {128: 189, 192: 317, 256: 569}[l]
    bits = 2 * l
    R = 1 << bits
    P = R - C
    nb = bits // 8
    Ls, Hs = [], []
    his = [0, 1, 2, (1 << 32) - 1, 1 << 32, 1 << (bits - 1), R - 1, R - 2] + [rnd.getrandbits(bits) for _ in range(40)]
    for K in (1, 3):
        for H in his:
            for delta in (-2, -1, 0, 1, 2, 1 << 31, (1 << 32) - 1, 1 << 32, (1 << 64) - 1):
                # want K*(L + C*H) = c*R + (R - c*C + delta) with 0 <= L < R: the top part c is
                # about K*C*H / R; try its neighbours and keep every c that gives a valid L
                base = K * C * H
                for c in range(max(0, base // R - 1), base // R + K + 2):
                    target = c * R + (R - c * C + delta)
                    target += (-target) % K                 # K | target (moves delta by < K)
                    L = (target - base) // K
                    if 0 <= L < R and (K * (L + C * H)) // R == c:
                        Ls.append(L); Hs.append(H)
    Ls += [R - 1, R - 1, 0, R - 1] + [rnd.getrandbits(bits) for _ in range(500)]
    Hs += [0, 1, R - 1, R - 1] + [rnd.getrandbits(bits) for _ in range(500)]
    assert len(Ls) > 600
    ta = dev(b"".join(x.to_bytes(nb, "little") for x in Ls))
    tb = dev(b"".join(x.to_bytes(nb, "little") for x in Hs))
    out = torch.empty_like(ta)
    for op, K in ((9, 1), (10, 3), (109, 1), (110, 3)):          # 109 / 110: the verification flavour (rare-branch ripple)
        code = eng.lib.bee2hip_debug_feL(ctypes.c_size_t(l), op, ctypes.c_void_p(ta.data_ptr()),
                                         ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_size_t(len(Ls)), None)
        assert code == 0
        torch.cuda.synchronize()
        raw = host(out)
        got = [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]
        want = [K * (a + R * b) % P for a, b in zip(Ls, Hs)]
        bad = [(i, hex(Ls[i]), hex(Hs[i])) for i in range(len(Ls)) if got[i] != want[i]]
        assert not bad, (l, K, len(bad), bad[:3])


@pytest.mark.parametrize("l", [128, 192, 256])
def test_field_ops_verification_flavour_rare_carry_branches(l):
    """the VtOps forms of add / sub / neg / mul / sqr / fold (debug ops 100 + op; what bign_main_kernel and bign_prep_kernel
    run): the second carry pass sits behind a wavefront-uniform branch taken with probability 2^-24 per lane on random
    data, so the operands are crafted to take it -- a + b = 2^(2l) + x with limb 0 of x within c of 2^32 and 0..N-1 limbs
    of ones above it (the carry ripples that far, and out of the top for a, b >= p), a - b = -(y) with limb 0 of
    2^(2l) - y below c over 0..N-1 zero limbs, folds that land within +-2 of 2^(32N) -- mixed with random lanes in the
    same wavefronts, and compared with Python integers AND with the straight-line (CtOps) form of the same operation."""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    rnd = random.Random(1000 + l)
    C = {128: 189, 192: 317, 256: 569}[l]
    bits, nb = 2 * l, l // 4
    R = 1 << bits
    P = R - C
    N = bits // 32
    A, B = [], []
    for j in range(N):                       # ripple length j limbs
        ones = ((1 << (32 * j)) - 1) << 32
        for k in list(range(1, 6)) + [C - 1, C, C + 1, C + 2]:
            for hi in (0, 1 << (32 * (j + 1)) if j + 1 < N else 0, rnd.getrandbits(bits) >> (32 * (j + 1)) << (32 * (j + 1))):
                x = (hi & (R - 1)) | ones | ((1 << 32) - k)                   # a + b = R + x
                a = rnd.randrange(x + 1, R) if x + 1 < R else R - 1
                b = R + x - a
                if 0 <= b < R:
                    A.append(a); B.append(b)
                y = R - ((hi & (R - 1) & ~((1 << (32 * (j + 1))) - 1)) | (k - 1))   # a - b = -y: t = R - y has limb 0 = k - 1 < c
                b2 = rnd.randrange(y, R) if y < R else R - 1
                a2 = b2 - y
                if 0 <= a2 < R:
                    A.append(a2); B.append(b2)
    zone = [P + i for i in range(C)]          # non-canonical operands: second wrap of an addition
    A += [R - 1, R - 1, P, R - 1, 0, 0, 1] + [rnd.choice(zone) for _ in range(64)] + [rnd.getrandbits(bits) for _ in range(2000)]
    B += [R - 1, P, P, 1, R - 1, P, R - 1] + [rnd.choice(zone) for _ in range(64)] + [rnd.getrandbits(bits) for _ in range(2000)]
    order = list(range(len(A)))
    rnd.shuffle(order)                        # rare lanes spread over the wavefronts
    A = [A[i] for i in order]; B = [B[i] for i in order]
    ta = dev(b"".join(x.to_bytes(nb, "little") for x in A))
    tb = dev(b"".join(x.to_bytes(nb, "little") for x in B))
    out = torch.empty_like(ta)

    def run(op):
        code = eng.lib.bee2hip_debug_feL(ctypes.c_size_t(l), op, ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()),
                                         ctypes.c_void_p(out.data_ptr()), ctypes.c_size_t(len(A)), None)
        assert code == 0
        torch.cuda.synchronize()
        raw = host(out)
        return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]
    ops = {0: lambda a, b: a * b % P, 1: lambda a, b: a * a % P, 2: lambda a, b: (a + b) % P, 3: lambda a, b: (a - b) % P,
           5: lambda a, b: 3 * a * b % P, 6: lambda a, b: 8 * a * a % P, 9: lambda a, b: (a + R * b) % P,
           10: lambda a, b: 3 * (a + R * b) % P}
    for op, f in ops.items():
        want = [f(a, b) for a, b in zip(A, B)]
        vt, ct = run(100 + op), run(op)
        bad = [(i, hex(A[i]), hex(B[i])) for i in range(len(A)) if vt[i] != want[i]]
        assert not bad, (l, op, len(bad), bad[:3])
        assert ct == want
    assert run(104) == [(-a) % P for a in A]
    # the point doubling in both flavours (x of 2P for Z = 1; any field values, on the curve or not -- same formulas)
    assert run(108) == run(8)


@pytest.mark.parametrize("l,path", [(192, 1), (192, 3), (256, 1), (256, 3)])
def test_bign_big_curves_both_kernel_sets(golden, l, path):
    """the wider curves: the r01 kernels (32-bit limbs, one lane per signature) and the quad kernel on 28- / 27-bit
    limbs (bign_fe29.hpp LZ<12>, LZ<16>), each FORCED over the base and edge fixtures and a 20 000-signature tiling with
    every 5th signature corrupted (quads are chosen unforced up to 2^14)"""
    eng = exp_engine()          # hooks of include/bee2hip_internal.h
    tune = eng.lib.bee2hip_internal_tune
    tune.restype = ctypes.c_uint32
    assert tune(2, path) == 0
    try:
        d = golden.bign_big[str(l)]
        oid = E.LEVEL_OID[l]
        no = l // 4
        cases = [dict(t, code=0, name="base") for t in d["base"]] + d["edge"]
        hs = b"".join(bytes.fromhex(c["hash"]) for c in cases)
        ss = b"".join(bytes.fromhex(c["sig"]) for c in cases)
        ps = b"".join(bytes.fromhex(c["pubkey"]) for c in cases)
        codes = torch.full((len(cases),), -1, dtype=torch.int32, device="cuda")
        eng.bignVerifyL_batch_dev(l, oid, dev(hs), dev(ss), dev(ps), codes)
        torch.cuda.synchronize()
        got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
        bad = [(c["name"], g, c["code"]) for c, g in zip(cases, got) if g != c["code"]]
        assert not bad, bad[:10]
        base = d["base"]
        n = 20_000
        reps = n // len(base) + 1
        H = np.frombuffer(b"".join(bytes.fromhex(t["hash"]) for t in base) * reps, dtype=np.uint8).copy()[: no * n]
        S = np.frombuffer(b"".join(bytes.fromhex(t["sig"]) for t in base) * reps, dtype=np.uint8).copy()[: (no + no // 2) * n]
        K = np.frombuffer(b"".join(bytes.fromhex(t["pubkey"]) for t in base) * reps, dtype=np.uint8).copy()[: 2 * no * n]
        S.reshape(n, no + no // 2)[::5, 3] ^= 0x10
        codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        eng.bignVerifyL_batch_dev(l, oid, torch.from_numpy(H).cuda(), torch.from_numpy(S).cuda(), torch.from_numpy(K).cuda(), codes)
        torch.cuda.synchronize()
        got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        want = np.zeros(n, dtype=np.int64)
        want[::5] = 510
        assert np.array_equal(got, want)
    finally:
        tune(2, 0)


@pytest.mark.parametrize("l", [192, 256])
def test_bign_big_curves_batch_and_dropin(orc, golden, l):
    eng = engine()
    d = golden.bign_big[str(l)]
    oid = E.LEVEL_OID[l]
    cases = [dict(t, code=0, name="base") for t in d["base"]] + d["edge"]
    hs = b"".join(bytes.fromhex(c["hash"]) for c in cases)
    ss = b"".join(bytes.fromhex(c["sig"]) for c in cases)
    ps = b"".join(bytes.fromhex(c["pubkey"]) for c in cases)
    codes = torch.full((len(cases),), -1, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_batch_dev(l, oid, dev(hs), dev(ss), dev(ps), codes)
    torch.cuda.synchronize()
    got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    bad = [(c["name"], g, c["code"]) for c, g in zip(cases, got) if g != c["code"]]
    assert not bad, bad[:10]
    assert {0, 505, 510} <= set(got)
    # drop-in facades and the generic bignVerify with the level's parameters
    params = eng.bignParamsStd(E.CURVE_NAME[l])
    for c in cases[:3] + d["edge"][:6]:
        h, s, p = (bytes.fromhex(c[x]) for x in ("hash", "sig", "pubkey"))
        assert eng.bignLVerify(l, h, s, p) == c["code"]
        assert eng.bignVerify(params, oid, h, s, p) == c["code"]
    code, host_codes = eng.bignVerify_batch(hs, ss, ps, oid_der=oid, params=params)
    assert code == 0 and host_codes == [c["code"] for c in cases]
    assert got == orc.verify_batch_l(l, oid, hs, ss, ps, nthreads=8)


@pytest.mark.parametrize("l", [128, 192, 256])
def test_bign_pubkey_val_batch_and_dropin(orc, golden, l):
    """bignPubkeyVal (bign_misc.c:319-365), the step before each verification in `sig vfy` (cmd_sig.c:463-478)"""
    eng = engine()
    cases = golden.bign_pubkey_val[str(l)]
    want = [c["code"] for c in cases]
    ps = b"".join(bytes.fromhex(c["pubkey"]) for c in cases)
    codes = torch.full((len(cases),), -1, dtype=torch.int32, device="cuda")
    eng.bignPubkeyValL_batch_dev(l, dev(ps), codes)
    torch.cuda.synchronize()
    got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    bad = [(c["name"], g, c["code"]) for c, g in zip(cases, got) if g != c["code"]]
    assert not bad, bad[:10]
    assert set(got) == {0, 505}
    params = eng.bignParamsStd(E.CURVE_NAME[l])
    code, host = eng.bignPubkeyVal_batch(ps, params)
    assert code == 0 and host == want
    for c in cases[:4] + cases[92:110]:
        pk = bytes.fromhex(c["pubkey"])
        assert eng.bignLPubkeyVal(l, pk) == c["code"], c["name"]
        assert eng.bignPubkeyVal(params, pk) == c["code"], c["name"]
    # a larger seeded batch against the oracle: the golden keys tiled, every third one corrupted
    rnd = random.Random(l)
    big = bytearray(ps * 40)
    n = len(big) // (l // 2)
    for i in range(0, n, 3):
        big[i * (l // 2) + rnd.randrange(l // 2)] ^= 1 << rnd.randrange(8)
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bignPubkeyValL_batch_dev(l, dev(bytes(big)), codes)
    torch.cuda.synchronize()
    assert [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()] == orc.pubkey_val_batch(l, bytes(big))
    # errors: wrong level, empty batch, bad parameters
    assert eng.lib.bee2hip_bignPubkeyValL_batch_dev(E._sz(100), None, E._sz(0), None, None) == 502
    assert eng.lib.bee2hip_bignPubkeyValL_batch_dev(E._sz(l), None, E._sz(0), None, None) == 0
    bad_params = eng.bignParamsStd(E.CURVE_NAME[l])
    bad_params.b[0] ^= 1
    # not one of the three standard curves: served by the general-curve kernels, where this key is off the curve
    # (tests/test_gpu_bign_generic.py holds the reference-generated fixtures for such parameter sets)
    assert eng.bignPubkeyVal(bad_params, bytes.fromhex(cases[0]["pubkey"])) == 505


@pytest.mark.parametrize("l,n", [(128, 2 * 32768 + 37), (128, 9 * 32768 - 5), (192, 2 * 65536 + 37), (256, 2 * 65536 + 3),
                                 (128, 2 ** 18 + 1), (192, 2 ** 18 + 3), (256, 2 ** 18 + 1),
                                 (128, 32767), (128, 32768), (128, 65535), (128, 65536), (128, 2 ** 18 - 1), (128, 2 ** 18),
                                 # round 2: the quad kernel's workgroup switch (2^13) and the kernel switches at 2^15 / 2^16
                                 (128, 3), (128, 17), (128, 8191), (128, 8192), (128, 8193), (128, 16385), (128, 32769),
                                 (128, 65537), (192, 8193), (192, 16384), (192, 16385), (256, 8192), (256, 16385)])
def test_bign_shared_inversion_groups_with_mixed_statuses(golden, l, n):
    """bign_inv_kernel shares one inversion between K signatures (K = n / 32768 resp. n / 65536, here 2, 4 and 8)
    and from 2^18 signatures on bign_prep_kernel normalises the tables of two signatures with one inversion.
    Batches whose groups mix genuine signatures with every edge case of the reference-generated fixtures
    (range errors that never reach the inversion, R = O, slow-path lanes, wrong signatures), sizes that
    are not multiples of K: each code must be the reference's.  On the 256-bit curve the sizes also sit on both sides
    of every kernel choice: one signature per quad up to 2^15 (64-thread workgroups up to 2^13), 29-bit limbs up to
    2^16, 32-bit limbs above."""
    eng = engine()
    if l == 128:
        hs, ss, ps = golden.bign_base_arrays()
        cases = [(hs[32 * i:32 * i + 32], ss[48 * i:48 * i + 48], ps[64 * i:64 * i + 64], 0) for i in range(len(hs) // 32)]
        cases += [(bytes.fromhex(e["hash"]), bytes.fromhex(e["sig"]), bytes.fromhex(e["pubkey"]), e["code"])
                  for e in golden.bign_edge]
    else:
        d = golden.bign_big[str(l)]
        cases = [(bytes.fromhex(t["hash"]), bytes.fromhex(t["sig"]), bytes.fromhex(t["pubkey"]), t.get("code", 0))
                 for t in d["base"] + d["edge"] * 3]
    rnd = random.Random(n)
    pick = [rnd.randrange(len(cases)) for _ in range(n)]
    H = b"".join(cases[i][0] for i in pick)
    S = b"".join(cases[i][1] for i in pick)
    P = b"".join(cases[i][2] for i in pick)
    want = np.array([cases[i][3] for i in pick], dtype=np.int64)
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_batch_dev(l, E.LEVEL_OID[l], dev(H), dev(S), dev(P), codes)
    torch.cuda.synchronize()
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (l, n, bad[:5], got[bad[:5]], want[bad[:5]])
    if n >= 1000:
        assert {0, 505, 510} <= set(int(x) for x in np.unique(got))


def test_bign_oid_lengths_every_alignment(golden):
    """The tail hashes oid || <x_R> || H with the OID's length shifting every later byte: genuine signatures
    under OIDs of 3..128 DER octets (all alignments mod 4, up to the staging limit), the reference as signer.
    Device batches per (curve, OID) and the generic drop-in."""
    eng = engine()
    groups = {}
    for c in golden.bign_oid_lengths:
        groups.setdefault((c["l"], c["oid"]), []).append(c)
    assert len(groups) >= 90
    for (l, oid_hex), cs in groups.items():
        oid = bytes.fromhex(oid_hex)
        cs = cs * 40                                        # several wavefronts
        hs = b"".join(bytes.fromhex(c["hash"]) for c in cs)
        ss = b"".join(bytes.fromhex(c["sig"]) for c in cs)
        ps = b"".join(bytes.fromhex(c["pubkey"]) for c in cs)
        codes = torch.full((len(cs),), -1, dtype=torch.int32, device="cuda")
        eng.bignVerifyL_batch_dev(l, oid, dev(hs), dev(ss), dev(ps), codes)
        torch.cuda.synchronize()
        got = [int(x) & 0xFFFFFFFF for x in codes.cpu().numpy()]
        assert got == [c["code"] for c in cs], (l, len(oid))
    params = eng.bignParamsStd(E.CURVE_NAME[128])
    for c in [c for c in golden.bign_oid_lengths if c["l"] == 128][:12]:
        assert eng.bignVerify(params, bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["sig"]),
                              bytes.fromhex(c["pubkey"])) == c["code"]


def test_bign_long_oids_prefix_hashed_once_per_batch(golden):
    """OIDs beyond the 128 octets a kernel stages (129 .. 4099 DER octets, every block alignment of the prefix): the whole
    32-byte blocks of the OID are absorbed once per batch and the per-signature kernels start from that state.  Device
    batches on all three curves, the generic drop-in, and -- for the valid entries -- the deterministic signature itself."""
    eng = engine()
    groups = {}
    for c in golden.bign_oid_long:
        groups.setdefault((c["l"], c["oid"]), []).append(c)
    assert len(groups) == 66
    for (l, oid_hex), cs in groups.items():
        oid = bytes.fromhex(oid_hex)
        cs = cs * 70
        hs = b"".join(bytes.fromhex(c["hash"]) for c in cs)
        ss = b"".join(bytes.fromhex(c["sig"]) for c in cs)
        ps = b"".join(bytes.fromhex(c["pubkey"]) for c in cs)
        codes = torch.full((len(cs),), -1, dtype=torch.int32, device="cuda")
        eng.bignVerifyL_batch_dev(l, oid, dev(hs), dev(ss), dev(ps), codes)
        torch.cuda.synchronize()
        got = [int(x) & 0xFFFFFFFF for x in codes.cpu().numpy()]
        assert got == [c["code"] for c in cs], (l, len(oid))
    for l in (128, 192, 256):
        params = eng.bignParamsStd(E.CURVE_NAME[l])
        for c in [c for c in golden.bign_oid_long if c["l"] == l]:
            oid, h, sg, pub = (bytes.fromhex(c[k]) for k in ("oid", "hash", "sig", "pubkey"))
            assert eng.bignVerify(params, oid, h, sg, pub) == c["code"]
            if c["code"] == 0:
                out = ctypes.create_string_buffer(len(sg))
                assert eng.lib.bignSign2(out, ctypes.byref(params), oid, ctypes.c_size_t(len(oid)), h, bytes.fromhex(c["privkey"]),
                                         None, ctypes.c_size_t(0)) == 0
                assert out.raw == sg, (l, len(oid))


def test_bign_big_batch_every_entry_against_the_oracle(orc, golden):
    """2^18 + 17 signatures on the 256-bit curve (shared inversions in prep and inv, big-table tail), HALF of them
    damaged by a random bit flip in the hash, s0, s1 or the public key -- each damaged key gives a table of its
    own -- and every single verdict compared with the oracle's."""
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    nb = len(hs) // 32
    n = (1 << 18) + 17
    reps = -(-n // nb)
    H = np.tile(np.frombuffer(hs, dtype=np.uint8), reps).reshape(-1, 32)[:n].copy()
    S = np.tile(np.frombuffer(ss, dtype=np.uint8), reps).reshape(-1, 48)[:n].copy()
    K = np.tile(np.frombuffer(ps, dtype=np.uint8), reps).reshape(-1, 64)[:n].copy()
    rng = np.random.default_rng(0x2B18)
    bad = rng.choice(n, n // 2, replace=False)
    kind = rng.integers(0, 4, bad.size)
    bit = (1 << rng.integers(0, 8, bad.size)).astype(np.uint8)
    for arr, k, width in ((H, 0, 32), (S, 1, 48), (K, 2, 64), (K, 3, 64)):
        sel = bad[kind == k]
        arr[sel, rng.integers(0, width, sel.size)] ^= bit[kind == k]
    K[rng.choice(n, 64, replace=False), :32] = 0xFF                  # x_Q >= p: never reaches the inversions
    S[rng.choice(n, 64, replace=False), 16:] = 0xFF                  # s1 >= q
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(H.reshape(-1)), dev(S.reshape(-1)), dev(K.reshape(-1)), codes)
    torch.cuda.synchronize()
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    want = np.array(orc.verify_batch(H.tobytes(), S.tobytes(), K.tobytes(), nthreads=64), dtype=np.int64)
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, (diff[:5], got[diff[:5]], want[diff[:5]])
    assert int((want == 0).sum()) >= n // 2 - 128 and {505, 510} <= set(np.unique(want).tolist())


def test_bign_device_batch_of_2pow22_signatures(orc, golden):
    """Sized for the card: 2^22 + 5 signatures resident in HBM in ONE call (16x BASELINE's configs[3]; scratch of the
    verification pipeline ~4.5 GB).  One entry in 64 damaged by a bit flip in s0 / s1 / the hash / the key: every untouched entry
    verifies, every damaged one does not, and 4096 of the damaged ones are compared with the oracle's codes."""
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    nb = len(hs) // 32
    n = (1 << 22) + 5
    reps = -(-n // nb)
    H = np.tile(np.frombuffer(hs, dtype=np.uint8), reps).reshape(-1, 32)[:n].copy()
    S = np.tile(np.frombuffer(ss, dtype=np.uint8), reps).reshape(-1, 48)[:n].copy()
    K = np.tile(np.frombuffer(ps, dtype=np.uint8), reps).reshape(-1, 64)[:n].copy()
    rng = np.random.default_rng(0x2B22)
    bad = np.sort(rng.choice(n, n // 64, replace=False))
    bad[:3] = (0, 1 << 18, n - 1)
    bad = np.unique(bad)
    kind = rng.integers(0, 3, bad.size)
    bit = (1 << rng.integers(0, 8, bad.size)).astype(np.uint8)
    for arr, k, width in ((H, 0, 32), (S, 1, 48), (K, 2, 64)):
        sel = bad[kind == k]
        arr[sel, rng.integers(0, width, sel.size)] ^= bit[kind == k]
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bign128Verify_batch_dev(dev(H.reshape(-1)), dev(S.reshape(-1)), dev(K.reshape(-1)), codes)
    torch.cuda.synchronize()
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    mask = np.zeros(n, dtype=bool)
    mask[bad] = True
    assert int((got[~mask] != 0).sum()) == 0
    assert int((got[mask] == 0).sum()) == 0
    sample = bad[rng.choice(bad.size, 4096, replace=False)]
    want = np.array(orc.verify_batch(H[sample].tobytes(), S[sample].tobytes(), K[sample].tobytes(), nthreads=32), dtype=np.int64)
    assert np.array_equal(got[sample], want)
