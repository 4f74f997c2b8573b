"""GPU: bignVerify / bignPubkeyVal on NON-STANDARD parameter sets (bign_generic_kernels.hip) against the fixtures the
reference produced (tests/golden/bign_generic.json) and against the Python restatement (tests/orc_generic.py) on
fresh random damage."""
import ctypes
import json
import os
import random

import pytest

import orc_generic as OG
from bee2_amd.engine import bign_params
from gpulib import engine

pytestmark = pytest.mark.gpu

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bign_generic.json")))


def mk(c):
    prm = bign_params()
    prm.l = c["l"]
    for f in ("p", "a", "b", "q", "yG"):
        raw = bytes.fromhex(c[f])
        ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
    return prm


def test_generic_verify_batches_match_the_reference():
    """every case of a curve in ONE bee2hip_bignVerify_batch call: valid signatures on isomorphic images of the standard
    curves (reference as signer) and on random-prime curves (no-wrap signatures the reference accepts), damaged variants"""
    eng = engine()
    for ci, c in enumerate(FIX["curves"]):
        prm = mk(c)
        cases = [x for x in FIX["cases"] if x["curve"] == ci]
        H = b"".join(bytes.fromhex(x["hash"]) for x in cases)
        S = b"".join(bytes.fromhex(x["sig"]) for x in cases)
        K = b"".join(bytes.fromhex(x["pubkey"]) for x in cases)
        code, got = eng.bignVerify_batch(H, S, K, oid_der=bytes.fromhex(cases[0]["oid"]), params=prm)
        assert code == 0, (ci, code)
        bad = [(x["name"], g, x["code"]) for x, g in zip(cases, got) if g != x["code"]]
        assert not bad, (ci, c["kind"], c["l"], bad[:5])
        assert 0 in got and 510 in got


def test_generic_verify_dropin_and_pubkey_val():
    eng = engine()
    for ci, c in enumerate(FIX["curves"]):
        prm = mk(c)
        cases = [x for x in FIX["cases"] if x["curve"] == ci][:3]
        for x in cases:
            got = eng.bignVerify(prm, bytes.fromhex(x["oid"]), bytes.fromhex(x["hash"]), bytes.fromhex(x["sig"]), bytes.fromhex(x["pubkey"]))
            assert got == x["code"], (ci, x["name"], got)
        pv = [x for x in FIX["pubkey_val"] if x["curve"] == ci]
        code, got = eng.bignPubkeyVal_batch(b"".join(bytes.fromhex(x["pubkey"]) for x in pv), prm)
        assert code == 0 and got == [x["code"] for x in pv], (ci, got)
        assert eng.bignPubkeyVal(prm, bytes.fromhex(pv[0]["pubkey"])) == pv[0]["code"]


def test_malformed_parameter_sets_report_the_reference_codes():
    eng = engine()
    for c in FIX["bad_params"]:
        prm = mk(c)
        h, s, k = (bytes.fromhex(c[x]) for x in ("hash", "sig", "pubkey"))
        assert eng.bignVerify(prm, bytes.fromhex(c["oid"]), h, s, k) == c["verify"], c["name"]
        assert eng.bignPubkeyVal(prm, k) == c["pubkey_val"], c["name"]
        # precedence as bignVerify: parameters (incl. what bignEcCreate rejects), then inputs, then the OID
        codes = (ctypes.c_uint32 * 1)()
        code = eng.lib.bee2hip_bignVerify_batch(ctypes.byref(prm), b"\x06\x01", ctypes.c_size_t(2), h, s, k, ctypes.c_size_t(1), codes)
        assert code == (c["verify"] if c["verify"] not in (0, 505, 510) else 301), c["name"]     # 301 = ERR_BAD_OID


def test_generic_verify_random_damage_vs_python_restatement(orc):
    """fresh corruptions of the fixture triples (the fixtures fix only a few): the Python restatement, pinned to the
    reference by tests/test_oracle_golden.py, is the checker"""
    eng = engine()
    rnd = random.Random(0x67656E)
    for ci, c in enumerate(FIX["curves"]):
        if c["l"] == 256 and c["kind"] == "rnd":
            continue                                            # the Python checker needs ~0.3 s per 512-bit case
        prm = mk(c)
        P = OG.Params.from_hex(c)
        no = c["l"] // 4
        good = [x for x in FIX["cases"] if x["curve"] == ci and x["name"] == "good"]
        H, S, K, want = b"", b"", b"", []
        for _ in range(12):
            x = rnd.choice(good)
            h, s, k = (bytearray.fromhex(x[f]) for f in ("hash", "sig", "pubkey"))
            r = rnd.randrange(5)
            if r == 1:
                s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
            elif r == 2:
                h[rnd.randrange(no)] ^= 1 << rnd.randrange(8)
            elif r == 3:
                k[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
            elif r == 4:
                s[no // 2:] = rnd.getrandbits(8 * no).to_bytes(no, "little")
            H += bytes(h); S += bytes(s); K += bytes(k)
            want.append(OG.verify(P, bytes.fromhex(x["oid"]), bytes(h), bytes(s), bytes(k), orc.belt_hash))
        code, got = eng.bignVerify_batch(H, S, K, oid_der=bytes.fromhex(good[0]["oid"]), params=prm)
        assert code == 0 and got == want, (ci, got, want)


SFIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bign_generic_sign.json")))


@pytest.mark.parametrize("which", range(len(SFIX)))
def test_generic_signing_side_matches_the_reference(which):
    """round 3: bignPubkeyCalc / bignKeypairGen / bignSign2 / bignSign on isomorphic images of the standard curves (a != -3;
    the constant-time general-curve ladder of bign_generic_kernels.hip) -- keys 0, 1, q - 1, q, 2^2l - 1, hashes at and beyond
    q, additional input up to 200 octets (the host-hashed theta path), a 203-octet OID, rejected rng draws, a malformed OID;
    every expected value from the reference (tools/make_golden_generic_sign.py).  What is signed here must also verify here."""
    eng = engine()
    ent = SFIX[which]
    c = FIX["curves"][ent["curve"]]
    P = mk(c)
    no = c["l"] // 4
    for x in ent["pubkey_calc"]:
        code, pub = eng.bignPubkeyCalc(P, bytes.fromhex(x["priv"]))
        assert code == x["code"], x["priv"]
        if code == 0:
            assert pub.hex() == x["pub"]
            assert eng.bignPubkeyVal(P, pub) == 0
    for x in ent["keypair_gen"]:
        rng = eng.rng_from_bytes(bytes.fromhex(x["rnd"]) + bytes(70 * no))
        code, priv, pub = eng.bignKeypairGen(P, rng)
        assert code == x["code"], x
        if code == 0:
            assert (priv.hex(), pub.hex(), rng.pos[0]) == (x["priv"], x["pub"], x["used"])
    for x in ent["sign2"]:
        t = None if x["t"] is None else bytes.fromhex(x["t"])
        oid, h, d = (bytes.fromhex(x[k]) for k in ("oid", "hash", "priv"))
        code, sig = eng.bignSign2(P, oid, h, d, t)
        assert code == x["code"], x
        if code == 0:
            assert sig.hex() == x["sig"], x
            assert eng.bignVerify(P, oid, h, sig, eng.bignPubkeyCalc(P, d)[1]) == 0
    for x in ent["sign"]:
        rng = eng.rng_from_bytes(bytes.fromhex(x["rnd"]))
        code, sig = eng.bignSign(P, bytes.fromhex(x["oid"]), bytes.fromhex(x["hash"]), bytes.fromhex(x["priv"]), rng)
        assert code == x["code"], x
        if code == 0:
            assert (sig.hex(), rng.pos[0]) == (x["sig"], x["used"])
        else:
            assert rng.pos[0] == 0


def test_generic_sign_batch_and_verify_roundtrip():
    """a batch of a few hundred deterministic signatures on a non-standard set through bee2hip_bignSign2_batch, every one
    verified by bee2hip_bignVerify_batch on the same set, plus refused keys in the middle of the batch"""
    eng = engine()
    c = FIX["curves"][SFIX[0]["curve"]]
    P = mk(c)
    no = c["l"] // 4
    q = int.from_bytes(bytes.fromhex(c["q"]), "little")
    rnd = random.Random(99)
    n = 200
    privs = [rnd.randrange(1, q).to_bytes(no, "little") for _ in range(n)]
    privs[17] = bytes(no)
    privs[150] = q.to_bytes(no, "little")
    hs = [rnd.randbytes(no) for _ in range(n)]
    oid = bytes.fromhex(SFIX[0]["sign2"][0]["oid"])
    code, sigs, codes = eng.bignSign2_batch(P, oid, b"".join(hs), b"".join(privs), None)
    assert code == 0
    assert [i for i in range(n) if codes[i] != 0] == [17, 150] and codes[17] == codes[150] == 504
    good = [i for i in range(n) if codes[i] == 0]
    pcode, pubs, pcodes = eng.bignPubkeyCalc_batch(P, b"".join(privs[i] for i in good))
    assert pcode == 0 and all(x == 0 for x in pcodes)
    sg = no + no // 2
    vcode, vcodes = eng.bignVerify_batch(b"".join(hs[i] for i in good), b"".join(sigs[sg * i: sg * i + sg] for i in good), pubs,
                                         oid_der=oid, params=P)
    assert vcode == 0 and all(x == 0 for x in vcodes)
