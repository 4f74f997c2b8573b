#!/usr/bin/env python3
"""Randomised differential campaign: libbee2hip (GPU) against the oracle on random sizes, keys, splits
and corruptions, every entry point family.  TEST INFRASTRUCTURE (uses oracle/): run on the GPU box,
   python tests/fuzz_gpu.py [seconds] [seed]
Prints one line per family with the number of cases, and stops at the first mismatch with a
reproducer (family, seed of the case)."""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bee2_amd  # noqa: E402
import goldenlib  # noqa: E402
import orclib  # noqa: E402
from bee2_amd.engine import LEVEL_OID  # noqa: E402

# FUZZ_LIB=exp: the experiments build (libbee2hip_exp.so), whose hooks force each kernel family at every size; default: the
# product library, dispatch by batch size as a caller gets it (the forcing calls below are then no-ops)
eng = bee2_amd.load_experiments() if os.environ.get("FUZZ_LIB", "product") == "exp" else bee2_amd.load()


def _tune(key, value):
    if eng.experiments:
        eng.lib.bee2hip_internal_tune(key, value)


eng.set_device(0)
orc = orclib.load()
G = goldenlib.Golden()


def dev(b):
    return torch.from_numpy(np.frombuffer(bytes(b) + bytes(16), dtype=np.uint8).copy()).cuda()[: len(b)]


def host(t):
    return t.cpu().numpy().tobytes()


def size(rnd, hi, edges=()):
    """mostly small, sometimes around a boundary, sometimes large"""
    r = rnd.random()
    if edges and r < 0.35:
        return max(0, rnd.choice(edges) + rnd.randrange(-2, 3))
    if r < 0.8:
        return rnd.randrange(0, min(hi, 2000))
    return rnd.randrange(0, hi)


def splits_of(rnd, n, align=1):
    out, left = [], n
    while left:
        s = min(left, align * rnd.choice((1, 2, 3, 7, 16, 33, 100, 1000)))
        out.append(s)
        left -= s
    return out or [0]


def f_bashF(rnd):
    n = size(rnd, 300_000, (64, 256, 1024, 65536))
    data = orc.fill(192 * n, rnd.getrandbits(32))
    t = dev(data)
    if n:
        eng.bashF_batch_dev(t)
    torch.cuda.synchronize()
    return host(t) == orc.bashF_batch(data, nthreads=8)


def f_ctr(rnd):
    n = size(rnd, 1 << 22, (16, 1024 * 16, 65536 * 16))
    key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
    msg = orc.fill(n, rnd.getrandbits(32))
    sp = splits_of(rnd, n) if n < 20000 else [n]
    want = orc.ctr(msg, key, iv)
    return eng.beltCTR_steps(msg, key, iv, sp)[0] == want and eng.beltCTR(msg, key, iv) == (0, want)


def f_mac_hash(rnd):
    n = size(rnd, 200_000, (16, 32, 128, 192))
    key = rnd.randbytes(rnd.choice((16, 24, 32)))
    msg = orc.fill(n, rnd.getrandbits(32))
    l = rnd.choice((128, 192, 256))
    ok = eng.beltMAC(msg, key) == (0, orc.mac(msg, key)) and eng.bashHash(l, msg) == orc.bashHash(l, msg)
    ok = ok and eng.beltHash(msg) == (0, orc.belt_hash(msg))
    if n < 5000:
        sp = splits_of(rnd, n)
        ok = ok and eng.beltMAC_steps(msg, key, sp) == (orc.mac(msg, key), True)
        ok = ok and eng.bashHash_steps(l, msg, sp) == (orc.bashHash(l, msg)[1], True)
        ok = ok and eng.beltHash_steps(msg, sp)[-1] == orc.belt_hash(msg)
    return ok


def f_modes(rnd):
    nb = max(1, size(rnd, 1 << 16, (1, 2, 64, 1024, 8192)))
    key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
    msg = orc.fill(16 * nb + (rnd.randrange(16) if rnd.random() < 0.5 else 0), rnd.getrandbits(32))
    blocks = msg[: 16 * nb]
    ok = True
    for decr in (False, True):
        d = "Decr" if decr else "Encr"
        ok = ok and eng.belt_mode("beltECB" + d, msg, key) == orc.ecb(msg, key, decr)
        ok = ok and eng.belt_mode("beltCBC" + d, msg, key, iv) == orc.cbc(msg, key, iv, decr)
        ok = ok and eng.belt_mode("beltBDE" + d, blocks, key, iv) == orc.bde(blocks, key, iv, decr)
        if 2 <= nb <= 600:
            ok = ok and eng.belt_mode("beltSDE" + d, blocks, key, iv) == orc.sde(blocks, key, iv, decr)
    if nb > 1:
        sp = splits_of(rnd, 16 * nb, align=16)
        decr = rnd.random() < 0.5
        ok = ok and eng.belt_mode_steps("BDE", decr, blocks, key, iv, sp) == orc.bde(blocks, key, iv, decr)[1]
    return ok


def f_aead(rnd):
    mode = rnd.choice(("DWP", "CHE"))
    nc, no = size(rnd, 1 << 20, (16, 1024 * 16, 64 * 1024 * 16)), size(rnd, 1 << 18, (16, 1024 * 16))
    key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
    crit, op = orc.fill(nc, rnd.getrandbits(32)), orc.fill(no, rnd.getrandbits(32))
    want = orc.dwp_wrap(crit, op, key, iv, mode)
    ok = eng.dwp_wrap(crit, op, key, iv, mode) == want
    mac = want[2] if rnd.random() < 0.7 else bytes([want[2][0] ^ 4]) + want[2][1:]
    ok = ok and eng.dwp_unwrap(want[1], op, mac, key, iv, mode) == orc.dwp_unwrap(want[1], op, mac, key, iv, mode)
    if nc + no < 6000:
        ops = [("I", p) for p in (op[i:i + 37] for i in range(0, no, 37))] + [("G",)]
        ops += [("E", crit[i:i + 29]) for i in range(0, nc, 29)]
        ops += [("A", want[1][i:i + 41]) for i in range(0, nc, 41)] + [("G",)]
        o, m, _ = eng.dwp_steps(key, iv, ops, mode)
        ok = ok and (o, m) == orc.dwp_steps(key, iv, ops, mode)
    return ok


_base = {128: None, 192: None, 256: None}


def _triples(l):
    if _base[l] is None:
        if l == 128:
            hs, ss, ps = G.bign_base_arrays()
            no = 32
            _base[l] = [(hs[no * i:no * i + no], ss[48 * i:48 * i + 48], ps[64 * i:64 * i + 64]) for i in range(len(hs) // no)]
        else:
            _base[l] = [tuple(bytes.fromhex(t[k]) for k in ("hash", "sig", "pubkey")) for t in G.bign_big[str(l)]["base"]]
    return _base[l]


_generic = None


def f_verify_generic(rnd):
    """a non-standard parameter set (general-curve kernels): fixture triples, fresh damage, the Python restatement as checker"""
    global _generic
    import json
    import orc_generic as OG
    from bee2_amd.engine import bign_params
    if _generic is None:
        _generic = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bign_generic.json")))
    ci = rnd.randrange(len(_generic["curves"]))
    c = _generic["curves"][ci]
    if c["l"] == 256:
        ci = 0
        c = _generic["curves"][0]
    prm = bign_params()
    prm.l = c["l"]
    for f in ("p", "a", "b", "q", "yG"):
        raw = bytes.fromhex(c[f])
        ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
    P = OG.Params.from_hex(c)
    no = c["l"] // 4
    good = [x for x in _generic["cases"] if x["curve"] == ci and x["name"] == "good"]
    H, S, K, want = b"", b"", b"", []
    for _ in range(rnd.randrange(1, 9)):
        x = rnd.choice(good)
        h, s, k = (bytearray.fromhex(x[f]) for f in ("hash", "sig", "pubkey"))
        r = rnd.randrange(6)
        if r == 1:
            s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
        elif r == 2:
            h[rnd.randrange(no)] ^= 1 << rnd.randrange(8)
        elif r == 3:
            k[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
        elif r == 4:
            s[no // 2:] = b"\xff" * no
        elif r == 5:
            k[:no] = b"\xff" * no
        H += bytes(h); S += bytes(s); K += bytes(k)
        want.append(OG.verify(P, bytes.fromhex(x["oid"]), bytes(h), bytes(s), bytes(k), orc.belt_hash))
    code, got = eng.bignVerify_batch(H, S, K, oid_der=bytes.fromhex(good[0]["oid"]), params=prm)
    if code != 0 or got != want:
        return False
    code, got = eng.bignPubkeyVal_batch(K, prm)
    return code == 0 and got == [OG.pubkey_val(P, K[2 * no * i:2 * no * (i + 1)]) for i in range(len(want))]


_std = {}


def _std_params(l):
    from bee2_amd.engine import CURVE_NAME
    if l not in _std:
        _std[l] = eng.bignParamsStd(CURVE_NAME[l])
    return _std[l]


def f_verify(rnd):
    if rnd.randrange(12) == 0:
        return f_verify_generic(rnd)
    l = rnd.choice((128, 128, 192, 256))
    base = _triples(l)
    n = max(1, size(rnd, 6000 if l == 128 else 800, (64, 256, 1024)))
    no = l // 4
    H, S, P = bytearray(), bytearray(), bytearray()
    for _ in range(n):
        h, s, p = (bytearray(x) for x in rnd.choice(base))
        k = rnd.randrange(10)
        if k == 1:
            s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
        elif k == 2:
            h[rnd.randrange(no)] ^= 1 << rnd.randrange(8)
        elif k == 3:
            p[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
        elif k == 4:
            s[no // 2:] = b"\xff" * no                      # s1 >= q
        elif k == 5:
            p[:no] = b"\xff" * no                           # x >= p
        elif k == 6:
            p[no:] = b"\xff" * no                           # y >= p
        elif k == 7:
            s[:no // 2] = bytes(no // 2)                    # s0 = 0
        H += h; S += s; P += p
    oid = LEVEL_OID[l]
    if rnd.randrange(4) == 0:                               # another OID length: every byte of the tail message moves
        k = rnd.choice((1, 2, 3, 4, 5, 8, 13, 20, 61, 126))
        oid = bytes([0x06, k] if k < 128 else [0x06, 0x81, k]) + bytes([0x2A] + [rnd.randrange(1, 128) for _ in range(k - 1)])
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    # which kernels walk the scalar multiplication is a matter of batch size (32-bit limbs / 29-bit limbs / one
    # signature per quad on the 256-bit curve): half of the cases force one of the three instead
    path = rnd.choice((0, 0, 0, 0, 1, 2, 0x43, 0x23, 0x83))
    _tune(2, path)
    try:
        eng.bignVerifyL_batch_dev(l, oid, dev(H), dev(S), dev(P), codes)
        torch.cuda.synchronize()
    finally:
        _tune(2, 0)
    got = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    if got != orc.verify_batch_l(l, oid, bytes(H), bytes(S), bytes(P), nthreads=16):
        return False
    # the drop-in symbol on a sample of the same cases, one call each: the host path (host_bign.hpp) in the default mode
    sg = no + no // 2
    prm = _std_params(l)
    pick = rnd.sample(range(n), min(n, 48))
    for i in pick:
        one = eng.bignVerify(prm, oid, bytes(H[no * i:no * (i + 1)]), bytes(S[sg * i:sg * (i + 1)]), bytes(P[2 * no * i:2 * no * (i + 1)]))
        if (one & 0xFFFFFFFF) != got[i]:
            return False
    # public-key validation over the same (partly damaged) keys
    eng.bignPubkeyValL_batch_dev(l, dev(P), codes)
    torch.cuda.synchronize()
    pv = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
    for i in pick:
        if (eng.bignLPubkeyVal(l, bytes(P[2 * no * i:2 * no * (i + 1)])) & 0xFFFFFFFF) != pv[i]:
            return False
    return pv == orc.pubkey_val_batch(l, bytes(P))


def f_ragged_mixed(rnd):
    nm = max(1, size(rnd, 3000, (64, 65, 1024)))
    msgs = [orc.fill(size(rnd, 20000, (0, 32, 64, 128, 192)), rnd.getrandbits(32)) for _ in range(nm)]
    alg = rnd.choice((0, 128, 192, 256))
    code, digs = eng.hash_ragged(alg, msgs)
    ok = code == 0 and all(d == (orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1]) for m, d in zip(msgs[:200], digs[:200]))
    ml = 16 * rnd.randrange(1, 300)
    n = max(1, size(rnd, 3000, (1024, 1025)))
    key = rnd.randbytes(32)
    data = orc.fill(ml * n, rnd.getrandbits(32))
    dig, tag = orc.mixed_batch(data, ml, key, nthreads=16)
    gd, gt = eng.bashHash_beltMAC_batch(data, ml, 256, key, n=n)
    return ok and gd == dig and gt == tag


def f_sign(rnd):
    """8f-4 tail: public keys, deterministic signatures (random t lengths, random OID lengths, bad keys mixed in),
    signatures with supplied one-time keys; every good signature must also verify on the device"""
    import ctypes
    from bee2_amd import engine as E
    l = rnd.choice((128, 128, 192, 256))
    no, sg = l // 4, 3 * l // 8
    P = eng.bignParamsStd(E.CURVE_NAME[l])
    q = int.from_bytes(bytes(P.q)[:no], "little")
    n = max(1, size(rnd, 1500 if l == 128 else 300, (1, 64, 256, 257)))
    privs = bytearray(orc.fill(no * n, rnd.getrandbits(32)))
    for i in range(n):
        k = rnd.randrange(40)
        if k == 0:
            privs[no * i: no * (i + 1)] = bytes(no)
        elif k == 1:
            privs[no * i: no * (i + 1)] = ((q + rnd.randrange(0, 5)) % (1 << (8 * no))).to_bytes(no, "little")
        elif k == 2:
            privs[no * i: no * (i + 1)] = rnd.choice((1, 2, q - 1, q - 2, q - 16, 15, 16, 1 << 128)).to_bytes(no, "little")
    privs = bytes(privs)
    hashes = bytearray(orc.fill(no * n, rnd.getrandbits(32)))
    if rnd.randrange(3) == 0:
        hashes[:no] = b"\xff" * no                               # H >= q
    hashes = bytes(hashes)
    oid = LEVEL_OID[l]
    if rnd.randrange(3) == 0:
        k = rnd.choice((1, 2, 3, 4, 5, 8, 13, 20, 61, 126))
        oid = bytes([0x06, k] if k < 128 else [0x06, 0x81, k]) + bytes([0x2A] + [rnd.randrange(1, 128) for _ in range(k - 1)])
    t = rnd.choice((None, b"", rnd.randbytes(rnd.randrange(1, 65)), rnd.randbytes(rnd.randrange(65, 200))))
    # k G by 1 (signed 6-bit or, 101, 4-bit windows) / 4 / 16 / 64 lanes per scalar, or as the product picks by batch size (0)
    _tune(10, rnd.choice((0, 0, 1, 1, 7, 8, 8, 72, 101, 102, 4, 16, 64)))
    try:
        return _sign_case(rnd, l, P, n, privs, hashes, oid, t)
    finally:
        _tune(10, 0)


def _sign_case(rnd, l, P, n, privs, hashes, oid, t):
    no, sg = l // 4, 3 * l // 8
    code, pubs, pc = eng.bignPubkeyCalc_batch(P, privs)
    if code:
        return False
    for i in range(n):
        w = orc.pubkey_calc(l, privs[no * i: no * (i + 1)])
        if w[0] != pc[i] or (w[0] == 0 and w[1] != pubs[2 * no * i: 2 * no * (i + 1)]):
            return False
    code, sigs, sc = eng.bignSign2_batch(P, oid, hashes, privs, t)
    if code:
        return False
    for i in range(n):
        w = orc.sign2(l, oid, hashes[no * i: no * (i + 1)], privs[no * i: no * (i + 1)], t)
        if w[0] != sc[i] or (w[0] == 0 and w[1] != sigs[sg * i: sg * (i + 1)]):
            return False
    good = [i for i in range(n) if sc[i] == 0]
    if good:
        code, vc = eng.bignVerify_batch(b"".join(hashes[no * i: no * (i + 1)] for i in good), b"".join(sigs[sg * i: sg * (i + 1)] for i in good),
                                        b"".join(pubs[2 * no * i: 2 * no * (i + 1)] for i in good), oid_der=oid, params=P)
        if code or any(vc):
            return False
    ks = orc.fill(no * n, rnd.getrandbits(32))
    code, sigs2, kc = eng.bignSignK_batch(P, oid, hashes, privs, ks)
    if code:
        return False
    for i in range(0, n, max(1, n // 40)):
        w = orc.sign_rnd(l, oid, hashes[no * i: no * (i + 1)], privs[no * i: no * (i + 1)], ks[no * i: no * (i + 1)])
        if w[0] != kc[i] or (w[0] == 0 and w[1] != sigs2[sg * i: sg * (i + 1)]):
            return False
    return True


def f_multi(rnd):
    """the *_multi host entries under a random logical device count (BEE2HIP_FAKE_DEVICES)"""
    import ctypes
    os.environ["BEE2HIP_FAKE_DEVICES"] = str(rnd.choice((1, 2, 3, 5, 8)))
    try:
        n = max(1, size(rnd, 50_000, (1, 7, 8, 9, 4096)))
        data = orc.fill(192 * n, rnd.getrandbits(32))
        buf = ctypes.create_string_buffer(data, len(data))
        if eng.lib.bee2hip_bashF_batch_multi(buf, ctypes.c_size_t(n), 0) or buf.raw != orc.bashF_batch(data, nthreads=8):
            return False
        key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        m = size(rnd, 1 << 21, (4095, 4096, 4097, 65536))
        msg = orc.fill(m, rnd.getrandbits(32))
        cut = rnd.choice((0, 0, 3, 16, 100)) if m > 200 else 0
        st = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
        eng.lib.beltCTRStart(st, key, ctypes.c_size_t(len(key)), iv)
        b = ctypes.create_string_buffer(msg, max(1, len(msg)))
        if cut and eng.lib.bee2hip_beltCTR_bulk(b, ctypes.c_size_t(cut), st):
            return False
        if eng.lib.bee2hip_beltCTR_bulk_multi(ctypes.byref(b, cut), ctypes.c_size_t(m - cut), st, 0):
            return False
        more = ctypes.create_string_buffer(bytes(40), 40)
        eng.lib.beltCTRStepE(more, ctypes.c_size_t(40), st)
        want = orc.ctr(msg + bytes(40), key, iv)
        return b.raw[:m] + more.raw == want
    finally:
        os.environ.pop("BEE2HIP_FAKE_DEVICES", None)


def f_sign_generic(rnd):
    """round 3: the signing side on a non-standard parameter set (isomorphic images of the standard curves: q is the group
    order).  Checker: the Python restatement of bignVerify (tests/orc_generic.py) must accept every signature the library
    makes, the library must verify them itself, and the public key must come out of bignPubkeyVal as valid; refused private
    keys keep their codes.  Long OIDs (up to ~600 octets) ride along on this family and on the standard curves (oracle)."""
    global _generic
    import json
    import orc_generic as OG
    from bee2_amd import engine as E
    from bee2_amd.engine import bign_params
    if _generic is None:
        _generic = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bign_generic.json")))
    iso = [i for i, c in enumerate(_generic["curves"]) if c["kind"] == "iso" and c["l"] <= 192]
    c = _generic["curves"][rnd.choice(iso)]
    prm = bign_params()
    prm.l = c["l"]
    for f in ("p", "a", "b", "q", "yG"):
        raw = bytes.fromhex(c[f])
        ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
    P = OG.Params.from_hex(c)
    l, no = c["l"], c["l"] // 4
    q = int.from_bytes(bytes.fromhex(c["q"]), "little")
    total = rnd.choice((11, 40, 129, 131, 160, 161, 200, 257, 258, 300, 600))
    hdr = 2 if total <= 129 else 3 if total <= 258 else 4
    body = total - hdr
    if (hdr == 2 and body >= 128) or (hdr == 3 and not 128 <= body < 256) or (hdr == 4 and body < 256):
        total, hdr, body = 11, 2, 9
    oid = (bytes([0x06, body]) if hdr == 2 else bytes([0x06, 0x81, body]) if hdr == 3 else bytes([0x06, 0x82, body >> 8, body & 255])) + \
        bytes([0x2A] + [rnd.randrange(1, 128) for _ in range(body - 1)])
    n = rnd.randrange(1, 6)
    privs = [rnd.randrange(1, q).to_bytes(no, "little") for _ in range(n)]
    bad = rnd.randrange(n) if rnd.randrange(3) == 0 else -1
    if bad >= 0:
        privs[bad] = rnd.choice((0, q, (1 << (8 * no)) - 1)).to_bytes(no, "little")
    hs = [rnd.randbytes(no) for _ in range(n)]
    t = rnd.choice((None, rnd.randbytes(rnd.randrange(1, 65)), rnd.randbytes(rnd.randrange(65, 150))))
    code, sigs, sc = eng.bignSign2_batch(prm, oid, b"".join(hs), b"".join(privs), t)
    if code:
        return False
    sg = no + no // 2
    for i in range(n):
        if i == bad:
            if sc[i] != 504:
                return False
            continue
        if sc[i] != 0:
            return False
        pc, pub = eng.bignPubkeyCalc(prm, privs[i])
        s_i = sigs[sg * i: sg * (i + 1)]
        if pc != 0 or eng.bignPubkeyVal(prm, pub) != 0 or OG.pubkey_val(P, pub) != 0:
            return False
        if OG.verify(P, oid, hs[i], s_i, pub, orc.belt_hash) != 0 or eng.bignVerify(prm, oid, hs[i], s_i, pub) != 0:
            return False
    # the same OID on the standard curve of the level: the C oracle reproduces the deterministic signature
    Ps = eng.bignParamsStd(E.CURVE_NAME[l])
    qs = int.from_bytes(bytes(Ps.q)[:no], "little")
    d = rnd.randrange(1, qs).to_bytes(no, "little")
    code, sig = eng.bignSign2(Ps, oid, hs[0], d, t)
    w = orc.sign2(l, oid, hs[0], d, t)
    if code != w[0] or (code == 0 and sig != w[1]):
        return False
    return code != 0 or eng.bignVerify(Ps, oid, hs[0], sig, eng.bignPubkeyCalc(Ps, d)[1]) == orc.verify_l(l, oid, hs[0], sig, orc.pubkey_calc(l, d)[1]) == 0


def f_onekey(rnd):
    """n signatures under ONE key (bee2hip_bignVerifyL_onekey_batch_dev / the host-pointer form): a random key -- sometimes a fixture's
    (Q = G among them), sometimes off the curve or >= p (the fallback) --, signatures of the fixtures and fresh ones made by the signing
    entry, random damage; the oracle's bignVerify on every entry.  Experiments build: the 8-bit / 16-bit table of the key forced."""
    l = rnd.choice((128, 128, 192, 256))
    no, sg = l // 4, 3 * l // 8
    oid = LEVEL_OID[l]
    base = _triples(l)
    n = max(1, size(rnd, 3000 if l == 128 else 500, (64, 256, 1024)))
    mode = rnd.randrange(5)
    H, S = bytearray(), bytearray()
    if mode == 0:                                               # a fixture's key with that fixture's and other fixtures' signatures
        h0, s0, pub = (bytes(x) for x in rnd.choice(base))
        for _ in range(n):
            h, s, p = rnd.choice(base) if rnd.randrange(3) else (h0, s0, pub)
            H += h; S += s
    else:                                                       # a fresh key, fresh signatures
        d = bytearray(orc.fill(no, rnd.randrange(1 << 30))); d[no - 1] &= 0x3F
        pub = orc.pubkey_calc(l, bytes(d))[1]
        hs = orc.fill(no * n, rnd.randrange(1 << 30))
        sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda"); c = torch.empty(n, dtype=torch.int32, device="cuda")
        eng.bignSign2L_batch_dev(l, oid, dev(hs), dev(bytes(d) * n), sigs, c)
        torch.cuda.synchronize()
        H += hs; S += host(sigs)
        if mode == 1:
            pub = bytearray(pub); pub[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8); pub = bytes(pub)     # off the curve
        elif mode == 2 and rnd.randrange(3) == 0:
            pub = (b"\xff" * no + pub[no:]) if rnd.randrange(2) else (pub[:no] + b"\xff" * no)              # a coordinate >= p
    for i in range(n):
        k = rnd.randrange(12)
        if k == 1:
            S[sg * i + rnd.randrange(sg)] ^= 1 << rnd.randrange(8)
        elif k == 2:
            H[no * i + rnd.randrange(no)] ^= 1 << rnd.randrange(8)
        elif k == 3:
            S[sg * i + no // 2: sg * (i + 1)] = b"\xff" * no     # s1 >= q
        elif k == 4:
            S[sg * i: sg * i + no // 2] = bytes(no // 2) if rnd.randrange(2) else b"\xff" * (no // 2)
    want = orc.verify_batch_l(l, oid, bytes(H), bytes(S), bytes(pub) * n, nthreads=16)
    _tune(20, rnd.choice((-1, 63, 0, 8)))
    _tune(22, rnd.choice((-1, 0, 1)))
    try:
        if rnd.randrange(3) == 0:
            code, got = eng.bignVerify_onekey_batch(bytes(H), bytes(S), bytes(pub), oid_der=oid, params=eng.bignParamsStd(bee2_amd.engine.CURVE_NAME[l]))
            return code == 0 and got == list(want)
        codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        for _ in range(rnd.choice((1, 1, 3))):                  # (a key crosses its table threshold between calls)
            eng.bignVerifyL_onekey_batch_dev(l, oid, dev(H), dev(S), bytes(pub), codes)
        torch.cuda.synchronize()
        return [int(x) & 0xFFFFFFFF for x in codes.cpu().numpy()] == list(want)
    finally:
        _tune(20, -1)
        _tune(22, -1)


def f_keyed(rnd):
    """n signatures of a few signers (bee2hip_bignVerifyL_keyed_batch_dev / the host-pointer form): fixture triples with their keys as the
    signers, some signatures handed to the wrong signer, junk keys, indices out of range, random damage; table forms and lane counts
    forced at random in the experiments build; the oracle's bignVerify with the signature's key on every entry"""
    l = rnd.choice((128, 128, 192, 256))
    no, sg = l // 4, 3 * l // 8
    oid = LEVEL_OID[l]
    base = _triples(l)
    n = max(1, size(rnd, 2500 if l == 128 else 400, (64, 256, 1024)))
    picks = [rnd.choice(base) for _ in range(n)]
    keys = sorted({p for _, _, p in rnd.sample(picks, min(n, rnd.choice((1, 2, 5, 17, 64))))} | {picks[0][2]})
    pos = {k: i for i, k in enumerate(keys)}
    idx, H, S = [], bytearray(), bytearray()
    for h, s_, p in picks:
        idx.append(pos.get(p, rnd.randrange(len(keys))) if rnd.randrange(8) else rnd.randrange(len(keys)))     # (the wrong signer now and then)
        H += h; S += s_
    for i in range(n):
        k = rnd.randrange(12)
        if k == 1:
            S[sg * i + rnd.randrange(sg)] ^= 1 << rnd.randrange(8)
        elif k == 2:
            H[no * i + rnd.randrange(no)] ^= 1 << rnd.randrange(8)
        elif k == 3:
            S[sg * i + no // 2: sg * (i + 1)] = b"\xff" * no
    keys = [bytearray(k) for k in keys]
    for k in keys:
        r = rnd.randrange(14)
        if r == 0:
            k[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)                   # off the curve
        elif r == 1:
            k[:no] = b"\xff" * no                                               # x >= p
    K = b"".join(bytes(k) for k in keys)
    want = orc.verify_batch_l(l, oid, bytes(H), bytes(S), b"".join(K[2 * no * i: 2 * no * (i + 1)] for i in idx), nthreads=16)
    idx2 = list(idx)
    for _ in range(rnd.choice((0, 0, 2))):
        j = rnd.randrange(n)
        idx2[j] = len(keys) + rnd.randrange(1 << 20)
        want[j] = 109                                                           # ERR_BAD_INPUT
    _tune(20, rnd.choice((-1, 63, 0, 8)))
    _tune(22, rnd.choice((-1, 0, 1)))
    try:
        if rnd.randrange(3) == 0:
            code, got = eng.bignVerify_keyed_batch(bytes(H), bytes(S), K, idx2, oid_der=oid, params=eng.bignParamsStd(bee2_amd.engine.CURVE_NAME[l]))
            return code == 0 and got == list(want)
        codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        ti = torch.tensor(idx2, dtype=torch.int64).to(torch.int32).cuda()
        for _ in range(rnd.choice((1, 1, 3))):
            eng.bignVerifyL_keyed_batch_dev(l, oid, dev(H), dev(S), K, ti, codes)
        torch.cuda.synchronize()
        return [int(x) & 0xFFFFFFFF for x in codes.cpu().numpy()] == list(want)
    finally:
        _tune(20, -1)
        _tune(22, -1)


FAMILIES = [("onekey", f_onekey), ("keyed", f_keyed), ("bashF", f_bashF), ("beltCTR", f_ctr), ("mac/hash", f_mac_hash), ("modes", f_modes), ("dwp/che", f_aead),
            ("verify", f_verify), ("ragged/mixed", f_ragged_mixed), ("sign", f_sign), ("multi", f_multi), ("sign-generic", f_sign_generic)]


def main(seconds, seed):
    global FAMILIES
    only = os.environ.get("FUZZ_FAMILIES")                  # e.g. FUZZ_FAMILIES=verify,sign
    if only:
        FAMILIES = [f for f in FAMILIES if f[0] in only.split(",")]
    counts = {n: 0 for n, _ in FAMILIES}
    t_end = time.time() + seconds
    case = 0
    while time.time() < t_end:
        name, fn = FAMILIES[case % len(FAMILIES)]
        cs = (seed << 20) + case
        if not fn(random.Random(cs)):
            print(f"MISMATCH family={name} case_seed={cs}  (reproduce: random.Random({cs}) into f_{name})")
            return 1
        counts[name] += 1
        case += 1
    for n, c in counts.items():
        print(f"{n:14s} {c:6d} cases ok")
    print(f"total {case} cases in {seconds} s, seed {seed}: no mismatch")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
