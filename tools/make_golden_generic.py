"""tests/golden/bign_generic.json: bignVerify / bignPubkeyVal on NON-STANDARD parameter sets, every expected code
taken from the reference itself (oracle/_ref/libbee2ref.so).  Build container only.

Curves:
  * "iso": images of the three standard curves under (x, y) -> (u^2 x, u^3 y): a' = a u^4, b' = b u^6, yG' = yG u^3,
    the same p and the same group order q -- a != -3, so the general doubling is exercised, and the reference can
    SIGN on them (bignKeypairGen / bignSign2 with these parameters);
  * "rnd": a random 2l-bit prime p = 3 (mod 4), random a, yG, b = yG^2, and a q that is NOT the group order (it
    cannot be computed here).  Valid signatures exist all the same: with a small private key d, a small hash H and a
    one-time key k > H + (s0 + 2^l) d below q, s1 = k - H - (s0 + 2^l) d involves no reduction mod q, and
    (s1 + H) G + (s0 + 2^l) Q = k G whatever the order of G is.  k G is computed here with textbook affine formulas;
    the reference's bignVerify then has to ACCEPT the result, or the script stops;
  * malformed parameter sets (a, b, yG >= p, p = 1 mod 4, even q, zero a / b, l = 96 ...) with the reference's code.
"""
import ctypes
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import refgen  # noqa: E402
import orc_generic as OG  # noqa: E402
from bee2_amd.engine import bign_params, LEVEL_OID  # noqa: E402

L = refgen.ref()
_sz = ctypes.c_size_t


def std(name):
    prm = bign_params()
    assert L.bignParamsStd(ctypes.byref(prm), name.encode()) == 0
    return prm


def mk(l, p, a, b, q, yG):
    prm = bign_params()
    prm.l = l
    no = l // 4
    for f, v in (("p", p), ("a", a), ("b", b), ("q", q), ("yG", yG)):
        raw = v if isinstance(v, (bytes, bytearray)) else v.to_bytes(no, "little")
        ctypes.memmove(getattr(prm, f), bytes(raw) + bytes(64 - len(raw)), 64)
    return prm


def hexp(prm):
    no = prm.l // 4
    return {"l": prm.l, **{f: bytes(getattr(prm, f))[:max(no, 1)].hex() if no else "" for f in ("p", "a", "b", "q", "yG")}}


def hexp_full(prm):        # malformed sets may carry junk beyond no octets: keep all 64
    return {"l": prm.l, **{f: bytes(getattr(prm, f)).hex() for f in ("p", "a", "b", "q", "yG")}}


def ref_verify(prm, oid, h, s, pub):
    return L.bignVerify(ctypes.byref(prm), oid, _sz(len(oid)), h, s, pub)


def ref_belt_hash(msg):
    out = ctypes.create_string_buffer(32)
    assert L.beltHash(out, bytes(msg), _sz(len(msg))) == 0
    return out.raw


def is_prime(n, rnd):
    if n % 2 == 0:
        return False
    for q in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for _ in range(24):
        x = pow(rnd.randrange(2, n - 1), d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


STD = {128: "1.2.112.0.2.0.34.101.45.3.1", 192: "1.2.112.0.2.0.34.101.45.3.2", 256: "1.2.112.0.2.0.34.101.45.3.3"}


def damage(rnd, l, prm, oid, h, s, pub, good_code):
    """variants of one triple with the reference's verdict for each"""
    no = l // 4
    out = []

    def put(name, hh, ss, pp):
        out.append({"name": name, "hash": bytes(hh).hex(), "sig": bytes(ss).hex(), "pubkey": bytes(pp).hex(),
                    "code": ref_verify(prm, oid, bytes(hh), bytes(ss), bytes(pp)) & 0xFFFFFFFF})
    put("good", h, s, pub)
    assert out[-1]["code"] == good_code, (out[-1], good_code)
    b = bytearray(s); b[rnd.randrange(no // 2)] ^= 1 << rnd.randrange(8); put("s0 bit", h, b, pub)
    b = bytearray(s); b[no // 2 + rnd.randrange(no)] ^= 1 << rnd.randrange(8); put("s1 bit", h, b, pub)
    b = bytearray(h); b[rnd.randrange(no)] ^= 1 << rnd.randrange(8); put("hash bit", b, s, pub)
    b = bytearray(s); b[no // 2:] = b"\xff" * no; put("s1 >= q", h, b, pub)
    b = bytearray(pub); b[:no] = b"\xff" * no; put("x >= p", h, s, b)
    b = bytearray(pub); b[no:] = b"\xff" * no; put("y >= p", h, s, b)
    b = bytearray(pub); b[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8); put("pubkey bit (off curve)", h, s, b)
    b = bytearray(s); b[:no // 2] = bytes(no // 2); put("s0 = 0", h, b, pub)
    return out


def main():
    rnd = random.Random(0x6E6572)
    rng = refgen.Combo(77)
    curves, cases, pubvals, badparams = [], [], [], []

    # ---- isomorphic images of the standard curves
    for l in (128, 192, 256):
        base = std(STD[l])
        no = l // 4
        p, a, b, q, yG = (OG.le(bytes(getattr(base, f))[:no]) for f in ("p", "a", "b", "q", "yG"))
        for rep in range(2 if l == 128 else 1):
            u = rnd.randrange(2, p)
            prm = mk(l, p, a * pow(u, 4, p) % p, b * pow(u, 6, p) % p, q, yG * pow(u, 3, p) % p)
            ci = len(curves)
            curves.append({"kind": "iso", **hexp(prm)})
            oid = bytes(LEVEL_OID[l])
            for k in range(6):
                priv = ctypes.create_string_buffer(no)
                pub = ctypes.create_string_buffer(2 * no)
                assert L.bignKeypairGen(priv, pub, ctypes.byref(prm), L.prngCOMBOStepR, rng.state) == 0
                h = rng.bytes(no)
                sig = ctypes.create_string_buffer(no + no // 2)
                assert L.bignSign2(sig, ctypes.byref(prm), oid, _sz(len(oid)), h, priv, None, _sz(0)) == 0
                for c in damage(rnd, l, prm, oid, h, sig.raw, pub.raw, 0) if k < 2 else damage(rnd, l, prm, oid, h, sig.raw, pub.raw, 0)[:1]:
                    cases.append({"curve": ci, "oid": oid.hex(), **c})
                pubvals.append({"curve": ci, "pubkey": pub.raw.hex(), "code": L.bignPubkeyVal(ctypes.byref(prm), pub.raw) & 0xFFFFFFFF})
                bad = bytearray(pub.raw); bad[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
                pubvals.append({"curve": ci, "pubkey": bytes(bad).hex(), "code": L.bignPubkeyVal(ctypes.byref(prm), bytes(bad)) & 0xFFFFFFFF})

    # ---- random primes, fake group order, no-wrap signatures
    for l in (128, 192, 256):
        no = l // 4
        for rep in range(2 if l == 128 else 1):
            while True:
                p = rnd.getrandbits(2 * l) | (1 << (2 * l - 1)) | 3
                if is_prime(p, rnd):
                    break
            a = rnd.randrange(1, p)
            yG = rnd.randrange(1, p)
            b = yG * yG % p
            q = rnd.getrandbits(2 * l) | (1 << (2 * l - 1)) | 1
            prm = mk(l, p, a, b, q, yG)
            assert OG.params_check(OG.Params.from_hex(hexp(prm))) == 0
            ci = len(curves)
            curves.append({"kind": "rnd", **hexp(prm)})
            oid = bytes(LEVEL_OID[l])
            G = (0, yG)
            for k in range(4):
                d = rnd.getrandbits(64) | 1
                Q = OG.mul(d, G, a, p)
                pub = Q[0].to_bytes(no, "little") + Q[1].to_bytes(no, "little")
                H = rnd.getrandbits(96)
                h = H.to_bytes(no, "little")
                kk = rnd.randrange(1 << (2 * l - 8), q >> 1) | (1 << (2 * l - 6))
                R = OG.mul(kk, G, a, p)
                t = ref_belt_hash(oid + R[0].to_bytes(no, "little") + h)
                s0 = t[:no // 2]
                s1 = kk - H - (OG.le(s0) + (1 << l)) * d
                assert 0 < s1 < q and s1 + H < q
                sig = s0 + s1.to_bytes(no, "little")
                for c in damage(rnd, l, prm, oid, h, sig, pub, 0) if k == 0 else damage(rnd, l, prm, oid, h, sig, pub, 0)[:1]:
                    cases.append({"curve": ci, "oid": oid.hex(), **c})
                pubvals.append({"curve": ci, "pubkey": pub.hex(), "code": L.bignPubkeyVal(ctypes.byref(prm), pub) & 0xFFFFFFFF})
                bad = bytearray(pub); bad[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
                pubvals.append({"curve": ci, "pubkey": bytes(bad).hex(), "code": L.bignPubkeyVal(ctypes.byref(prm), bytes(bad)) & 0xFFFFFFFF})

    # ---- malformed parameter sets (a genuine triple of the standard curve rides along: the code comes from the parameters)
    base = std(STD[128])
    tr = refgen.make_triples(1, 5)[0]
    oid = bytes(LEVEL_OID[128])
    p = OG.le(bytes(base.p)[:32])

    def bp(name, **ch):
        prm = std(STD[128])
        for f, v in ch.items():
            if f == "l":
                prm.l = v
            else:
                raw = v if isinstance(v, (bytes, bytearray)) else v.to_bytes(64, "little")
                ctypes.memmove(getattr(prm, f), bytes(raw) + bytes(64 - len(raw)), 64)
        badparams.append({"name": name, **hexp_full(prm), "hash": tr[0].hex(), "sig": tr[1].hex(), "pubkey": tr[2].hex(), "oid": oid.hex(),
                          "verify": ref_verify(prm, oid, tr[0], tr[1], tr[2]) & 0xFFFFFFFF,
                          "pubkey_val": L.bignPubkeyVal(ctypes.byref(prm), tr[2]) & 0xFFFFFFFF})
    bp("a = p", a=p)
    bp("a = p + 5", a=p + 5)
    bp("b = p", b=p)
    bp("yG = p", yG=p)
    bp("yG = 2^256 - 1", yG=(1 << 256) - 1)
    bp("a = 0", a=0)
    bp("b = 0", b=0)
    bp("p = 1 mod 4", p=p - 2)
    bp("p even", p=p - 1)
    bp("q even", q=OG.le(bytes(base.q)[:32]) - 1)
    bp("p short (top bit clear)", p=p >> 1 | 3)
    bp("q short", q=OG.le(bytes(base.q)[:32]) >> 1 | 1)
    bp("junk beyond p", p=bytes(base.p)[:32] + b"\x01" + bytes(31))
    bp("junk beyond yG", yG=bytes(base.yG)[:32] + b"\x01" + bytes(31))
    bp("l = 96", l=96)
    bp("l = 100", l=100)
    bp("l = 0", l=0)
    bp("l = 512", l=512)
    bp("a = 1 (valid parameters, another curve)", a=1)
    bp("yG = 1 (base point off the curve: still accepted)", yG=1)

    # cross-check with the Python restatement before writing
    PP = [OG.Params.from_hex(c) for c in curves]
    for c in cases:
        got = OG.verify(PP[c["curve"]], bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["sig"]),
                        bytes.fromhex(c["pubkey"]), ref_belt_hash)
        assert got == c["code"], (c["name"], got, c["code"])
    for c in pubvals:
        assert OG.pubkey_val(PP[c["curve"]], bytes.fromhex(c["pubkey"])) == c["code"]
    out = {"curves": curves, "cases": cases, "pubkey_val": pubvals, "bad_params": badparams}
    path = os.path.join(ROOT, "tests", "golden", "bign_generic.json")
    json.dump(out, open(path, "w"), indent=0)
    from collections import Counter
    print(len(curves), "curves,", len(cases), "verify cases", Counter(c["code"] for c in cases), len(pubvals), "pubkey cases",
          Counter(c["code"] for c in pubvals), len(badparams), "malformed sets",
          Counter((c["verify"], c["pubkey_val"]) for c in badparams))


if __name__ == "__main__":
    main()
