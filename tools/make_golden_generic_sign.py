#!/usr/bin/env python3
"""tests/golden/bign_generic_sign.json -- bignPubkeyCalc / bignKeypairGen / bignSign / bignSign2 on NON-STANDARD parameter
sets (round 3), every expected value produced by the REFERENCE (oracle/_ref/libbee2ref.so).  Build container only.

Curves: the "iso" sets of tests/golden/bign_generic.json (tools/make_golden_generic.py) -- images of the standard curves
under (x, y) -> (u^2 x, u^3 y): a != -3, the same p, and q IS the order of G, which the signing side needs (the
reference's regular scalar recodings use k -> q - k and the like; on a set whose q is not the group order their output is
an artefact of the recoding, not k G, and there is nothing to be on a par with).  rng-driven functions replay a recorded
byte stream through a gen_i callback, as tools/make_golden_sign.py does."""
import ctypes
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import refgen  # noqa: E402
import make_golden_sign as MS  # noqa: E402
from bee2_amd.engine import bign_params, LEVEL_OID  # noqa: E402

L = refgen.ref()
_sz = ctypes.c_size_t


def mkparams(c):
    prm = bign_params()
    prm.l = c["l"]
    for f in ("p", "a", "b", "q", "yG"):
        raw = bytes.fromhex(c[f])
        ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
    return prm


def le(x, n):
    return (x % (1 << (8 * n))).to_bytes(n, "little")


def main():
    rnd = random.Random(0x6E73)
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "bign_generic.json")))
    out = []
    for ci, c in enumerate(G["curves"]):
        if c["kind"] != "iso":
            continue
        prm = mkparams(c)
        l = c["l"]
        no = l // 4
        q = int.from_bytes(bytes.fromhex(c["q"]), "little")
        p = int.from_bytes(bytes.fromhex(c["p"]), "little")
        oid = bytes(LEVEL_OID[l])
        ent = {"curve": ci, "pubkey_calc": [], "keypair_gen": [], "sign2": [], "sign": []}
        privs = [le(rnd.randrange(1, q), no) for _ in range(6)]
        for d in privs + [le(v, no) for v in (0, 1, 2, q - 1, q, q + 1, (1 << (8 * no)) - 1)]:
            pub = ctypes.create_string_buffer(2 * no)
            code = L.bignPubkeyCalc(pub, ctypes.byref(prm), d) & 0xFFFFFFFF
            ent["pubkey_calc"].append({"priv": d.hex(), "code": code, "pub": pub.raw.hex() if code == 0 else ""})
        for s in (le(rnd.randrange(1, q), no), bytes(no) + le(rnd.randrange(1, q), no), le(p, no) + le(rnd.randrange(1, q), no), le(q - 1, no)):
            priv = ctypes.create_string_buffer(no)
            pub = ctypes.create_string_buffer(2 * no)
            cb = MS.replay(s)
            code = L.bignKeypairGen(priv, pub, ctypes.byref(prm), cb, None) & 0xFFFFFFFF
            ent["keypair_gen"].append({"rnd": s.hex(), "code": code, "priv": priv.raw.hex(), "pub": pub.raw.hex(), "used": cb.pos[0]})
        hs = [rnd.randbytes(no) for _ in range(8)]
        ts = [None, b"\x01", rnd.randbytes(23), rnd.randbytes(64), rnd.randbytes(65), rnd.randbytes(200)]
        long_oid = bytes([0x06, 0x81, 200, 0x2A]) + bytes(rnd.randrange(1, 128) for _ in range(199))
        cases = [(hs[i], privs[i], ts[i], oid) for i in range(6)]
        cases += [(le((1 << (8 * no)) - 1, no), privs[0], None, oid), (le(q, no), privs[1], b"x", oid), (bytes(no), privs[2], None, oid),
                  (hs[6], le(1, no), None, oid), (hs[7], le(q - 1, no), None, oid), (hs[0], le(0, no), None, oid), (hs[1], le(q, no), None, oid),
                  (hs[2], privs[3], None, bytes.fromhex("06022A03")), (hs[3], privs[4], b"abc", long_oid), (hs[4], privs[5], None, b"\x07\x01\x00")]
        for h, d, t, o in cases:
            sig = ctypes.create_string_buffer(no + no // 2)
            code = L.bignSign2(sig, ctypes.byref(prm), o, _sz(len(o)), h, d, t, _sz(len(t) if t else 0)) & 0xFFFFFFFF
            if code == 0:          # the reference verifies what it signed
                pub = ctypes.create_string_buffer(2 * no)
                assert L.bignPubkeyCalc(pub, ctypes.byref(prm), d) == 0
                assert L.bignVerify(ctypes.byref(prm), o, _sz(len(o)), h, sig.raw, pub.raw) == 0
            ent["sign2"].append({"oid": o.hex(), "hash": h.hex(), "priv": d.hex(), "t": None if t is None else t.hex(), "code": code,
                                 "sig": sig.raw.hex() if code == 0 else ""})
        for i in range(4):
            s = le(rnd.randrange(1, q), no)
            if i == 2:
                s = bytes(no) + le(q, no) + s                 # two rejected draws first
            d = privs[i] if i != 3 else le(q, no)             # bad key: the generator must stay untouched
            sig = ctypes.create_string_buffer(no + no // 2)
            cb = MS.replay(s)
            code = L.bignSign(sig, ctypes.byref(prm), oid, _sz(len(oid)), hs[i], d, cb, None) & 0xFFFFFFFF
            ent["sign"].append({"oid": oid.hex(), "hash": hs[i].hex(), "priv": d.hex(), "rnd": s.hex(), "code": code,
                                "sig": sig.raw.hex() if code == 0 else "", "used": cb.pos[0]})
        out.append(ent)
        print(f"curve {ci} (l = {l}): pubkey_calc {[x['code'] for x in ent['pubkey_calc']]}, keypair {[x['code'] for x in ent['keypair_gen']]}, "
              f"sign2 {[x['code'] for x in ent['sign2']]}, sign {[x['code'] for x in ent['sign']]}")
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "bign_generic_sign.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
