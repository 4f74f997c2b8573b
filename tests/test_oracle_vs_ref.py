"""CPU tests, build container only: the oracle restatement against the REFERENCE itself
(oracle/_ref/libbee2ref.so = agievich/bee2 compiled by oracle/Makefile) on seeded random
inputs.  Skipped where _ref is absent.  tests/pin_oracle.py runs the same comparison at
>= 1e5 items per primitive."""
import ctypes
import os
import random

import pytest

import refgen

pytestmark = pytest.mark.skipif(not refgen.have_ref(), reason="oracle/_ref not built")
_sz = ctypes.c_size_t


def test_bashF_100k_states(orc):
    L = refgen.ref()
    n = 100_000
    data = orc.fill(192 * n, 0xBA5F)
    ref = ctypes.create_string_buffer(data, len(data))
    base = ctypes.addressof(ref)
    for i in range(n):
        L.bashF(ctypes.c_void_p(base + 192 * i), None)
    assert orc.bashF_batch(data, nthreads=4) == ref.raw


def test_belt_ctr_100k_blocks(orc):
    L = refgen.ref()
    n = 100_000
    data = orc.fill(16 * n + 5, 0xBE17)
    H = orc.beltH()
    out = ctypes.create_string_buffer(len(data))
    assert L.beltCTR(out, data, _sz(len(data)), H[128:160], _sz(32), H[192:208]) == 0
    assert orc.ctr(data, H[128:160], H[192:208]) == out.raw
    # block-indexed form used by the GPU sharding: E_K(ctr0 + first + i + 1)
    import numpy as np
    key_w, ctr0_w = orc.ctr_start(H[128:160], H[192:208])
    arr = np.frombuffer(data[16 * 1000: 16 * 3000], dtype=np.uint8).copy()
    orc.ctr_blocks_np(arr, key_w, ctr0_w, first=1000, nthreads=3)
    assert arr.tobytes() == out.raw[16 * 1000: 16 * 3000]


def test_belt_mac_hash_random(orc):
    L = refgen.ref()
    rnd = random.Random(5)
    for _ in range(2000):
        n = rnd.choice((0, 1, 15, 16, 17, 32, 33, 64, 75, 100, 256, 1000))
        msg, key = rnd.randbytes(n), rnd.randbytes(rnd.choice((16, 24, 32)))
        a = ctypes.create_string_buffer(8)
        L.beltMAC(a, msg, _sz(n), key, _sz(len(key)))
        assert a.raw == orc.mac(msg, key)
        a = ctypes.create_string_buffer(32)
        L.beltHash(a, msg, _sz(n))
        assert a.raw == orc.belt_hash(msg)


def test_bash_hash_random(orc):
    L = refgen.ref()
    rnd = random.Random(6)
    for _ in range(1500):
        l = rnd.choice(range(16, 257, 16))
        n = rnd.choice((0, 1, 63, 64, 65, 95, 96, 127, 128, 129, 191, 192, 193, 500, 4096))
        msg = rnd.randbytes(n)
        a = ctypes.create_string_buffer(l // 4)
        assert L.bashHash(a, _sz(l), msg, _sz(n)) == 0
        assert a.raw == orc.bashHash(l, msg)[1]


def test_verify_random_and_corrupted(orc):
    rnd = random.Random(7)
    triples = refgen.make_triples(1500, 0x1234)
    seen = set()
    for i, (h, s, p) in enumerate(triples):
        kind = i % 6
        h, s, p = bytearray(h), bytearray(s), bytearray(p)
        if kind == 1:
            s[rnd.randrange(16)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            s[16 + rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        elif kind == 3:
            h[rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        elif kind == 4:
            p[rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        elif kind == 5:
            p[31] |= 0xFF; p[30] = 0xFF; p[24:30] = b"\xff" * 6
        want = refgen.verify(bytes(h), bytes(s), bytes(p))
        assert orc.verify(h, s, p) == want, (i, kind)
        seen.add(want)
    assert seen >= {0, 510}


def test_verify_big_curves_random_and_corrupted(orc):
    from bee2_amd.engine import LEVEL_OID
    rnd = random.Random(8)
    for l in (192, 256):
        seen = set()
        for i, (h, s, p) in enumerate(refgen.make_triples_l(l, 200, 0x55 + l)):
            h, s, p = bytearray(h), bytearray(s), bytearray(p)
            kind = i % 5
            if kind == 1:
                s[rnd.randrange(len(s))] ^= 1 << rnd.randrange(8)
            elif kind == 2:
                h[rnd.randrange(len(h))] ^= 1 << rnd.randrange(8)
            elif kind == 3:
                p[rnd.randrange(len(p))] ^= 1 << rnd.randrange(8)
            elif kind == 4:
                p[l // 4 - 1] = 0xFF; p[l // 4 - 2] = 0xFF; p[1:l // 4 - 2] = b"\xff" * (l // 4 - 3)
            want = refgen.verify_l(l, bytes(h), bytes(s), bytes(p))
            assert orc.verify_l(l, LEVEL_OID[l], h, s, p) == want, (l, i, kind)
            seen.add(want)
        assert seen >= {0, 510}


def test_pubkey_val_random_and_corrupted(orc):
    rnd = random.Random(11)
    L = refgen.ref()
    for l in (128, 192, 256):
        no = l // 4
        fn = getattr(L, f"bign{l}PubkeyVal")
        seen = set()
        for i in range(300):
            pub = bytearray(refgen.pubkey_calc_l(l, rnd.randbytes(no - 1) + b"\x00"))
            kind = i % 4
            if kind == 1:
                pub[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
            elif kind == 2:
                pub[1:no] = b"\xff" * (no - 1)                    # x in [p - 255, 2^(8 no))
            elif kind == 3:
                pub = bytearray(rnd.randbytes(2 * no))
            want = fn(bytes(pub))
            assert orc.pubkey_val(l, pub) == want, (l, i, kind)
            seen.add(want)
        assert seen == {0, 505}


def test_belt_bde_random(orc):
    """8f-1 belt-bde: one-shots of the reference on whole-block messages of every key size"""
    L = refgen.ref()
    rnd = random.Random(5)
    for _ in range(200):
        nb = rnd.choice((1, 2, 3, 63, 64, 65, 127, 128, 129, rnd.randrange(1, 800)))
        msg, key, iv = rnd.randbytes(16 * nb), rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        for fn, decr in (("beltBDEEncr", False), ("beltBDEDecr", True)):
            out = ctypes.create_string_buffer(len(msg))
            assert getattr(L, fn)(out, msg, _sz(len(msg)), key, _sz(len(key)), iv) == 0
            assert orc.bde(msg, key, iv, decr) == (0, out.raw), (fn, nb)


def test_belt_sde_wbl_random(orc):
    """8f-1 belt-sde and the whole-block belt-wbl under it: one-shots / Step{E,D} of the reference"""
    L = refgen.ref()
    L.beltWBL_keep.restype = _sz
    rnd = random.Random(31)
    for _ in range(150):
        nb = rnd.choice((2, 3, 4, 5, 6, 7, 8, 31, 32, 33, rnd.randrange(2, 300)))
        msg, key, iv = rnd.randbytes(16 * nb), rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        for fn, decr in (("beltSDEEncr", False), ("beltSDEDecr", True)):
            out = ctypes.create_string_buffer(len(msg))
            assert getattr(L, fn)(out, msg, _sz(len(msg)), key, _sz(len(key)), iv) == 0
            assert orc.sde(msg, key, iv, decr) == (0, out.raw), (fn, nb)
        st = ctypes.create_string_buffer(L.beltWBL_keep())
        L.beltWBLStart(st, key, _sz(len(key)))
        for step, decr in ((L.beltWBLStepE, False), (L.beltWBLStepD, True)):
            b = ctypes.create_string_buffer(msg, len(msg))
            step(b, _sz(len(msg)), st)
            assert orc.wbl(msg, key, decr) == (0, b.raw[: len(msg)]), (nb, decr)


@pytest.mark.parametrize("mode", ["DWP", "CHE"])
def test_belt_dwp_che_step_sequences_with_midstream_tags(orc, mode):
    """8f-2 belt-dwp / belt-che: the same randomly cut Step{I,E,A,G} sequence through the reference's state and
    the oracle's -- every tag taken mid-stream and the ciphertext must agree; then Wrap / Unwrap incl. bad mac"""
    L = refgen.ref()
    getattr(L, f"belt{mode}_keep").restype = _sz
    rnd = random.Random(19)
    R = lambda name: getattr(L, f"belt{mode}{name}")          # the reference's function of this mode

    def cut(b):
        parts = []
        while b:
            k = rnd.choice((1, 3, 7, 15, 16, 17, 33, 64))
            parts.append(b[:k])
            b = b[k:]
        return parts
    for _ in range(400):
        crit = rnd.randbytes(rnd.choice((0, 1, 7, 15, 16, 17, 31, 32, 33, 100, rnd.randrange(0, 600))))
        op = rnd.randbytes(rnd.choice((0, 1, 15, 16, 17, 32, 47, rnd.randrange(0, 300))))
        key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        st = ctypes.create_string_buffer(getattr(L, f"belt{mode}_keep")())
        R("Start")(st, key, _sz(len(key)), iv)
        ops, refout, refmacs = [], b"", []

        def tag():
            m = ctypes.create_string_buffer(8)
            R("StepG")(m, st)
            refmacs.append(m.raw)
            ops.append(("G",))
        for part in cut(op):
            ops.append(("I", part))
            R("StepI")(part, _sz(len(part)), st)
            if rnd.random() < 0.3:
                tag()
        for part in cut(crit):
            b = ctypes.create_string_buffer(part, len(part))
            R("StepE")(b, _sz(len(part)), st)
            refout += b.raw[: len(part)]
            ops.append(("E", part))
        for part in cut(refout):
            ops.append(("A", part))
            R("StepA")(part, _sz(len(part)), st)
            if rnd.random() < 0.3:
                tag()
        tag()
        assert orc.dwp_steps(key, iv, ops, mode) == (refout, refmacs)
        d, m = ctypes.create_string_buffer(max(len(crit), 1)), ctypes.create_string_buffer(8)
        assert R("Wrap")(d, m, crit, _sz(len(crit)), op, _sz(len(op)), key, _sz(len(key)), iv) == 0
        assert (d.raw[: len(crit)], m.raw) == (refout, refmacs[-1])
        assert orc.dwp_wrap(crit, op, key, iv, mode) == (0, refout, m.raw)
        bad = bytes([m.raw[0] ^ 1]) + m.raw[1:]
        d2 = ctypes.create_string_buffer(max(len(crit), 1))
        for mac in (m.raw, bad):
            rc = R("Unwrap")(d2, refout, _sz(len(refout)), op, _sz(len(op)), mac, key, _sz(len(key)), iv)
            oc, od = orc.dwp_unwrap(refout, op, mac, key, iv, mode)
            assert rc == oc == (0 if mac == m.raw else 511)
            if rc == 0:
                assert od == d2.raw[: len(crit)] == crit


def test_sign2_pubkey_calc_random(orc):
    """8f-4 tail: the oracle's bignSign2 / bignPubkeyCalc against the reference on random keys, hashes and
    additional inputs, all three curves; every reference signature verifies under the reference"""
    L = refgen.ref()
    rnd = random.Random(0x5164)
    oid = {128: bytes.fromhex("06092A7000020022651F51"), 192: bytes.fromhex("06092A7000020022654D0C"),
           256: bytes.fromhex("06092A7000020022654D0D")}
    for l in (128, 192, 256):
        no, sg, pk = refgen.SIZES[l]
        rng = refgen.Combo(4000 + l)
        for it in range(200 if l == 128 else 60):
            priv, pub = refgen.keypair_l(l, rng)
            assert orc.pubkey_calc(l, priv) == (0, pub)
            h = rng.bytes(no)
            t = rnd.randbytes(rnd.randrange(0, 80)) if it % 3 else None
            want = ctypes.create_string_buffer(sg)
            assert getattr(L, f"bign{l}Sign2")(want, h, priv, t, _sz(len(t) if t else 0)) == 0
            assert orc.sign2(l, oid[l], h, priv, t) == (0, want.raw)
            assert refgen.verify_l(l, h, want.raw, pub) == 0


def test_generic_parameter_fixtures_are_what_the_reference_says():
    """tests/golden/bign_generic.json replayed through the compiled reference: every verdict (verification, key
    validation, malformed parameter sets) must still be the reference's -- the fixture file cannot drift from it"""
    import ctypes
    import json
    from bee2_amd.engine import bign_params
    L = refgen.ref()
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bign_generic.json")))

    def mk(c):
        prm = bign_params()
        prm.l = c["l"]
        for f in ("p", "a", "b", "q", "yG"):
            raw = bytes.fromhex(c[f])
            ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
        return prm
    P = [mk(c) for c in d["curves"]]
    for c in d["cases"]:
        oid = bytes.fromhex(c["oid"])
        got = L.bignVerify(ctypes.byref(P[c["curve"]]), oid, ctypes.c_size_t(len(oid)), bytes.fromhex(c["hash"]),
                           bytes.fromhex(c["sig"]), bytes.fromhex(c["pubkey"])) & 0xFFFFFFFF
        assert got == c["code"], (c["curve"], c["name"])
    for c in d["pubkey_val"]:
        assert L.bignPubkeyVal(ctypes.byref(P[c["curve"]]), bytes.fromhex(c["pubkey"])) & 0xFFFFFFFF == c["code"]
    for c in d["bad_params"]:
        prm = mk(c)
        oid = bytes.fromhex(c["oid"])
        h, s, k = (bytes.fromhex(c[x]) for x in ("hash", "sig", "pubkey"))
        assert L.bignVerify(ctypes.byref(prm), oid, ctypes.c_size_t(len(oid)), h, s, k) & 0xFFFFFFFF == c["verify"], c["name"]
        assert L.bignPubkeyVal(ctypes.byref(prm), k) & 0xFFFFFFFF == c["pubkey_val"], c["name"]
