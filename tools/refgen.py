"""Drive the REFERENCE (oracle/_ref/libbee2ref.so, built from /root/reference by
oracle/Makefile) to produce signatures / known answers.  Build-container only:
used by tools/make_golden.py and by the `not gpu` oracle tests when _ref exists.
Test infrastructure, never imported by the product package."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libbee2ref.so")
REF_AVX512_SO = os.path.join(ROOT, "oracle", "_ref", "libbee2ref_avx512.so")

_sz = ctypes.c_size_t


def have_ref():
    return os.path.exists(REF_SO)


_lib = None


def ref():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(REF_SO)
        _lib.beltH.restype = ctypes.POINTER(ctypes.c_ubyte)
        _lib.prngCOMBO_keep.restype = _sz
    return _lib


def beltH():
    p = ref().beltH()
    return bytes(p[i] for i in range(256))


class Combo:
    """bee2 prngCOMBO (src/core/prng.c:39-67) as a gen_i source."""

    def __init__(self, seed):
        L = ref()
        self.state = ctypes.create_string_buffer(L.prngCOMBO_keep())
        L.prngCOMBOStart(self.state, ctypes.c_uint32(seed))

    def bytes(self, n):
        buf = ctypes.create_string_buffer(n)
        ref().prngCOMBOStepR(buf, _sz(n), self.state)
        return buf.raw


SIZES = {128: (32, 48, 64), 192: (48, 72, 96), 256: (64, 96, 128)}     # hash/priv, sig, pubkey octets


def keypair_l(l, rng):
    L = ref()
    no, _, pk = SIZES[l]
    priv = ctypes.create_string_buffer(no)
    pub = ctypes.create_string_buffer(pk)
    code = getattr(L, f"bign{l}KeypairGen")(priv, pub, L.prngCOMBOStepR, rng.state)
    assert code == 0, code
    return priv.raw, pub.raw


def sign2_l(l, h, priv):
    sig = ctypes.create_string_buffer(SIZES[l][1])
    code = getattr(ref(), f"bign{l}Sign2")(sig, h, priv, None, _sz(0))
    assert code == 0, code
    return sig.raw


def verify_l(l, h, sig, pub):
    return getattr(ref(), f"bign{l}Verify")(h, sig, pub)


def pubkey_calc_l(l, priv):
    pub = ctypes.create_string_buffer(SIZES[l][2])
    assert getattr(ref(), f"bign{l}PubkeyCalc")(pub, priv) == 0
    return pub.raw


def make_triples_l(l, n, seed, nkeys=16):
    rng = Combo(seed)
    keys = [keypair_l(l, rng) for _ in range(nkeys)]
    out = []
    for i in range(n):
        priv, pub = keys[i % nkeys]
        h = rng.bytes(SIZES[l][0])
        out.append((h, sign2_l(l, h, priv), pub))
    return out


def keypair(rng):
    L = ref()
    priv = ctypes.create_string_buffer(32)
    pub = ctypes.create_string_buffer(64)
    code = L.bign128KeypairGen(priv, pub, L.prngCOMBOStepR, rng.state)
    assert code == 0, code
    return priv.raw, pub.raw


def sign2(hash32, priv):
    """deterministic bign128Sign2 (bign_sign.c:140-245), t = empty"""
    sig = ctypes.create_string_buffer(48)
    code = ref().bign128Sign2(sig, hash32, priv, None, _sz(0))
    assert code == 0, code
    return sig.raw


def verify(hash32, sig, pub):
    return ref().bign128Verify(hash32, sig, pub)


def make_triples(n, seed, nkeys=64):
    """n genuine (hash, sig, pubkey) triples: nkeys keypairs, COMBO-random hashes."""
    rng = Combo(seed)
    keys = [keypair(rng) for _ in range(nkeys)]
    out = []
    for i in range(n):
        priv, pub = keys[i % nkeys]
        h = rng.bytes(32)
        out.append((h, sign2(h, priv), pub))
    return out
