"""TEST INFRASTRUCTURE: pure-Python restatement of bignVerify / bignPubkeyVal over an ARBITRARY parameter set
(bign_sign.c:268-361, bign_misc.c:319-365, bign_params.c:244-280, bign_ec.c:29-80), big integers and textbook affine
formulas; belt-hash comes from the C oracle (orclib).  Pinned by tests/test_oracle_golden.py against
tests/golden/bign_generic.json, which the reference itself produced (tools/make_golden_generic.py)."""

ERR_OK, ERR_BAD_INPUT, ERR_NOT_IMPLEMENTED = 0, 109, 119
ERR_BAD_PARAMS, ERR_BAD_PUBKEY, ERR_BAD_SIG = 502, 505, 510


def le(b):
    return int.from_bytes(bytes(b), "little")


class Params:
    def __init__(self, l, p, a, b, q, yG):
        self.l, self.p, self.a, self.b, self.q, self.yG = l, bytes(p), bytes(a), bytes(b), bytes(q), bytes(yG)

    @classmethod
    def from_hex(cls, d):
        return cls(d["l"], *(bytes.fromhex(d[k]) for k in ("p", "a", "b", "q", "yG")))


def params_check(P):
    """bignParamsCheck followed by what bignEcCreate rejects; the 64-octet fields arrive zero-padded"""
    l = P.l
    if 2 * l % 64:
        return ERR_NOT_IMPLEMENTED
    no = 2 * l // 8
    if no == 0 or no > 64:
        return ERR_BAD_PARAMS
    pad = lambda x: bytes(x) + bytes(64 - len(x))
    p, a, b, q, yG = (pad(x) for x in (P.p, P.a, P.b, P.q, P.yG))
    ok = (p[0] % 4 == 3 and q[0] % 2 == 1 and p[no - 1] >= 128 and q[no - 1] >= 128 and not any(p[no:])
          and any(a[:no]) and any(b[:no]) and not any(a[no:]) and not any(b[no:]) and not any(q[no:]) and not any(yG[no:]))
    if not ok:
        return ERR_BAD_PARAMS
    if l % 64:
        return ERR_NOT_IMPLEMENTED
    if l not in (128, 192, 256):
        return ERR_BAD_PARAMS
    pi = le(p[:no])
    if le(a[:no]) >= pi or le(b[:no]) >= pi or le(yG[:no]) >= pi:      # qrFrom in ecpCreateJ / ecGroupCreate
        return ERR_BAD_PARAMS
    return ERR_OK


def _add(P1, P2, a, p):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1 + a) * pow(2 * y1, p - 2, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, p - 2, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def mul(k, P, a, p):
    R = None
    for i in reversed(range(k.bit_length())):
        R = _add(R, R, a, p)
        if (k >> i) & 1:
            R = _add(R, P, a, p)
    return R


def verify(P, oid_der, h, sig, pub, belt_hash):
    code = params_check(P)
    if code:
        return code
    l = P.l
    no = l // 4
    p, a, q, yG = le(P.p[:no]), le(P.a[:no]), le(P.q[:no]), le(P.yG[:no])
    x, y = le(pub[:no]), le(pub[no:2 * no])
    if x >= p or y >= p:
        return ERR_BAD_PUBKEY
    s0, s1 = le(sig[:no // 2]), le(sig[no // 2:no // 2 + no])
    if s1 >= q:
        return ERR_BAD_SIG
    H = le(h[:no])
    if H >= q:
        H -= q
    u = (s1 + H) % q
    v = s0 + (1 << l)
    R = _add(mul(u, (0, yG), a, p), mul(v, (x, y), a, p), a, p)
    if R is None:
        return ERR_BAD_SIG
    t = belt_hash(bytes(oid_der) + R[0].to_bytes(no, "little") + bytes(h[:no]))
    return ERR_OK if t[:no // 2] == bytes(sig[:no // 2]) else ERR_BAD_SIG


def pubkey_val(P, pub):
    code = params_check(P)
    if code:
        return code
    no = P.l // 4
    p, a, b = le(P.p[:no]), le(P.a[:no]), le(P.b[:no])
    x, y = le(pub[:no]), le(pub[no:2 * no])
    if x >= p or y >= p:
        return ERR_BAD_PUBKEY
    return ERR_OK if (y * y - (x * x * x + a * x + b)) % p == 0 else ERR_BAD_PUBKEY
