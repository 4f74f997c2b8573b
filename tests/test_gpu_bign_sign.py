"""-m gpu: SURVEY 8f-4, second half -- bignPubkeyCalc / bignKeypairGen / bignSign / bignSign2 and their
bign128 / bign192 / bign256 facades through the C ABI (mirrors test/crypto/bign_test.c:303-456 and
bign128_test.c), plus the batch entry points.  Expected values: tests/golden/bign_sign.json (the reference's
answers, tools/make_golden_sign.py) and the oracle on seeded random inputs; signatures are deterministic
(bignSign2) or driven by a replayed rng stream, so every comparison is bit-exact."""
import threading

import numpy as np
import pytest
import torch

import orclib
from bee2_amd import engine as E
from gpulib import dev, engine, exp_engine, host

pytestmark = pytest.mark.gpu


def _params(eng, l):
    return eng.bignParamsStd(E.CURVE_NAME[l])


def test_stb_annex_G_vectors(golden):
    """G.1 (key pair from the rng's first draw), G.2 / G.3 (bignSign), G.6 / G.7 (bignSign2 without / with t)"""
    eng = engine()
    k = golden.bign_sign["stb"]
    oid = bytes.fromhex(k["oid"])
    P = _params(eng, 128)
    g1 = k["G1"]
    rng = eng.rng_from_bytes(bytes.fromhex(g1["rnd"]))
    code, priv, pub = eng.bignKeypairGen(P, rng)
    assert (code, priv.hex(), pub.hex(), rng.pos[0]) == (0, g1["priv"], g1["pub"], 32)
    assert eng.bignPubkeyCalc(P, priv) == (0, pub)
    assert eng.bignLPubkeyCalc(128, priv) == (0, pub)
    assert eng.bignPubkeyVal(P, pub) == 0
    for name in ("G2", "G3"):
        c = k[name]
        h = bytes.fromhex(c["hash"])
        rng = eng.rng_from_bytes(bytes.fromhex(c["rnd"]))
        code, sig = eng.bignSign(P, oid, h, priv, rng)
        assert (code, sig.hex(), rng.pos[0]) == (0, c["sig"], 32), name
        assert eng.bignVerify(P, oid, h, sig, pub) == 0
        rng = eng.rng_from_bytes(bytes.fromhex(c["rnd"]))
        assert eng.bignLSign(128, h, priv, rng) == (0, sig)
    for name in ("G6", "G7"):
        c = k[name]
        h = bytes.fromhex(c["hash"])
        t = None if c["t"] is None else bytes.fromhex(c["t"])
        code, sig = eng.bignSign2(P, oid, h, priv, t)
        assert (code, sig.hex()) == (0, c["sig"]), name
        assert eng.bignLSign2(128, h, priv, t) == (0, sig)
        assert eng.bign128Verify(h, sig, pub) == 0


@pytest.mark.parametrize("l", [128, 192, 256])
def test_golden_cases_dropin(golden, l, mulbase):
    """every case of the fixture through the drop-in functions: keys 0, 1, q-1, q-2, q, 2^2l - 1; hashes at and
    beyond q; additional input of 0..300 octets; rejected rng draws; foreign, long and malformed OIDs"""
    eng = mulbase
    P = _params(eng, l)
    L = golden.bign_sign[str(l)]
    no = l // 4
    for c in L["pubkey_calc"]:
        code, pub = eng.bignPubkeyCalc(P, bytes.fromhex(c["priv"]))
        assert code == c["code"], c["priv"]
        if code == 0:
            assert pub.hex() == c["pub"]
        assert eng.bignLPubkeyCalc(l, bytes.fromhex(c["priv"]))[0] == c["code"]
    for c in L["keypair_gen"]:
        rng = eng.rng_from_bytes(bytes.fromhex(c["rnd"]) + bytes(70 * no))
        code, priv, pub = eng.bignKeypairGen(P, rng)
        if not c["defined"]:                          # draw in [q, p): the reference's result is not d G (see the generator)
            assert rng.pos[0] == c["used"]
            continue
        assert code == c["code"], c
        if code == 0:
            assert (priv.hex(), pub.hex(), rng.pos[0]) == (c["priv"], c["pub"], c["used"])
            assert eng.bignPubkeyVal(P, pub) == 0
    for c in L["sign2"]:
        t = None if c["t"] is None else bytes.fromhex(c["t"])
        code, sig = eng.bignSign2(P, bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]), t)
        assert code == c["code"], c
        if code == 0:
            assert sig.hex() == c["sig"], c
    for c in L["sign"]:
        rng = eng.rng_from_bytes(bytes.fromhex(c["rnd"]))
        code, sig = eng.bignSign(P, bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]), rng)
        assert code == c["code"], c
        if code == 0:
            assert (sig.hex(), rng.pos[0]) == (c["sig"], c["used"])
        else:
            assert rng.pos[0] == 0                    # a bad private key must not consume the generator


@pytest.fixture(params=["auto", "gpu", 1, 7, 72, 8, 101, 102, 4, 16, 64],
                ids=lambda v: {"auto": "product_library_auto_single_calls_on_the_host", "gpu": "product_library_forced_gpu",
                               7: "1_lane_7bit_windows_looked_up_in_LDS", 72: "1_lane_7bit_LDS_other_coordinates", 8: "1_lane_8bit_windows_LDS_16_copies", 101: "1_lane_4bit_windows",
                               102: "1_lane_6bit_complete_additions"}.get(v, f"{v}_lanes_per_scalar"))
def mulbase(request):
    """Who computes k G.  "auto": the PRODUCT library as a caller gets it -- ONE key pair / signature through a drop-in symbol
    on the calling core in constant-time host arithmetic (bee2_amd/csrc/host_bign_ct.hpp), batches on the GPU with the kernel
    picked by batch size; "gpu": the product library under BEE2HIP_FORCE=gpu semantics (every secret in the kernels).  The
    numbers: one GPU form FORCED at every size through the hook of the experiments build (libbee2hip_exp.so, also forced to the
    GPU): one lane per scalar (bign_mulbase_ct_kernel: signed 6-bit windows and Jacobian mixed additions; 7 = signed 7-bit windows with the
    entry looked up in LDS, bign_mulbase_lds_kernel: the throughput form of the 256-bit curve from 2^18 scalars on -- the other curves fall back to 1; 102 =
    the same windows with complete additions, 101 = the round-2 form on unsigned 4-bit windows) or 4 / 16 / 64 lanes per scalar
    (bign_mulbase_coop_kernel).  Yields the engine to use; defaults restored."""
    if request.param in ("auto", "gpu"):
        eng = engine()
        was = eng.lib.bee2hip_path_policy(1 if request.param == "gpu" else 0)
        before = [eng.lib.bee2hip_path_count(i) for i in range(3)]
        yield eng
        after = [eng.lib.bee2hip_path_count(i) for i in range(3)]
        eng.lib.bee2hip_path_policy(was)
        if request.param == "gpu":
            assert after[0] == before[0], "host path taken under the forced GPU policy"
        assert after[2] == before[2]
        return
    eng = exp_engine()
    was = eng.lib.bee2hip_path_policy(1)
    eng.lib.bee2hip_internal_tune(10, request.param)
    yield eng
    eng.lib.bee2hip_internal_tune(10, 0)
    eng.lib.bee2hip_path_policy(was)


@pytest.mark.parametrize("l", [128, 192, 256])
def test_batch_vs_oracle_and_verify_roundtrip(orc, l, mulbase):
    """host batch API on seeded random items with bad keys mixed in: codes and outputs per item as the oracle,
    outputs of refused items untouched; then every signature verifies under its public key on the device"""
    eng = mulbase
    P = _params(eng, l)
    no, sg = l // 4, 3 * l // 8
    oid = E.LEVEL_OID[l]
    q = int.from_bytes(bytes(P.q)[:no], "little")
    n = 777 if l == 128 else 200
    privs = bytearray(orc.fill(no * n, 0x5164 + l))
    for i in range(0, n, 97):
        privs[no * i: no * (i + 1)] = bytes(no) if i % 2 else ((q + i) % (1 << (8 * no))).to_bytes(no, "little")
    privs = bytes(privs)
    hashes = orc.fill(no * n, 0xAA + l)
    code, pubs, codes = eng.bignPubkeyCalc_batch(P, privs)
    assert code == 0
    want_pub = [orc.pubkey_calc(l, privs[no * i: no * (i + 1)]) for i in range(n)]
    assert codes == [w[0] for w in want_pub]
    assert all(w[0] or w[1] == pubs[2 * no * i: 2 * no * (i + 1)] for i, w in enumerate(want_pub))
    assert all(w[0] == 0 or pubs[2 * no * i: 2 * no * (i + 1)] == bytes(2 * no) for i, w in enumerate(want_pub))
    for t in (None, b"\x05" * 17):
        code, sigs, scodes = eng.bignSign2_batch(P, oid, hashes, privs, t)
        assert code == 0
        want = [orc.sign2(l, oid, hashes[no * i: no * (i + 1)], privs[no * i: no * (i + 1)], t) for i in range(n)]
        assert scodes == [w[0] for w in want]
        assert all(w[0] or w[1] == sigs[sg * i: sg * (i + 1)] for i, w in enumerate(want))
    good = [i for i in range(n) if codes[i] == 0]
    vh = b"".join(hashes[no * i: no * (i + 1)] for i in good)
    vs = b"".join(sigs[sg * i: sg * (i + 1)] for i in good)
    vp = b"".join(pubs[2 * no * i: 2 * no * (i + 1)] for i in good)
    code, vcodes = eng.bignVerify_batch(vh, vs, vp, oid_der=oid, params=P)
    assert code == 0 and vcodes == [0] * len(good)
    # one-time keys supplied (bignSign after its rng): valid, zero and >= q
    ks = bytearray(orc.fill(no * n, 0x4B + l))
    ks[0:no] = bytes(no)
    ks[no: 2 * no] = q.to_bytes(no, "little")
    code, sigs2, kcodes = eng.bignSignK_batch(P, oid, hashes, privs, bytes(ks))
    assert code == 0
    for i in range(n):
        w = orc.sign_rnd(l, oid, hashes[no * i: no * (i + 1)], privs[no * i: no * (i + 1)], bytes(ks[no * i: no * (i + 1)]))
        assert kcodes[i] == w[0], i
        if w[0] == 0:
            assert sigs2[sg * i: sg * (i + 1)] == w[1]


@pytest.mark.parametrize("l", [128, 192, 256])
def test_base_point_multiples_of_special_scalars(orc, l, mulbase):
    """d G for d = 1, 2, 15, 16, 2^(4 j), q - 1, q - 2 and digit patterns that leave most windows empty or full; d = 0 and
    d >= q are refused with the reference's code; one-time key k = q - 1 and k = 1 sign like the oracle"""
    eng = mulbase
    P = _params(eng, l)
    no, sg = l // 4, 3 * l // 8
    q = int.from_bytes(bytes(P.q)[:no], "little")
    ds = [1, 2, 15, 16, 17, q - 1, q - 2, q >> 1, 0, q, q + 1, (1 << (8 * no)) - 1,
          int("f0" * no, 16) % q, int("0f" * no, 16) % q, int("01" * no, 16), 1 << (8 * no - 5)]
    ds += [1 << (4 * j) for j in range(1, 2 * no, 7)] + [15 << (4 * j) for j in range(0, 2 * no - 1, 5)]
    # signed 6-bit windows: every window 32 (digit -32 + carry), 31 (largest positive), 33, 63 (carry chains), mixtures
    bits = 8 * no
    rep = lambda pat: sum(v << (6 * i) for i, v in enumerate((pat * (bits // 6 + 2))[: bits // 6 + 1])) % (1 << (bits - 1))
    ds += [rep([32]), rep([31]), rep([33]), rep([63]), rep([31, 32]), rep([32, 63, 0]), rep([63, 63, 31, 0, 32]), rep([0, 0, 32])]
    privs = b"".join(d.to_bytes(no, "little") for d in ds)
    code, pubs, codes = eng.bignPubkeyCalc_batch(P, privs)
    assert code == 0
    for i, d in enumerate(ds):
        w = orc.pubkey_calc(l, privs[no * i: no * (i + 1)])
        assert codes[i] == w[0], hex(d)
        assert pubs[2 * no * i: 2 * no * (i + 1)] == (w[1] if w[0] == 0 else bytes(2 * no)), hex(d)
    oid = E.LEVEL_OID[l]
    good = [d for d in ds if 0 < d < q]
    hashes = orc.fill(no * len(good), 0x77 + l)
    dd = b"".join((q - 1 - i).to_bytes(no, "little") for i in range(len(good)))
    kk = b"".join(d.to_bytes(no, "little") for d in good)
    code, sigs, kcodes = eng.bignSignK_batch(P, oid, hashes, dd, kk)
    assert code == 0
    for i in range(len(good)):
        w = orc.sign_rnd(l, oid, hashes[no * i: no * (i + 1)], dd[no * i: no * (i + 1)], kk[no * i: no * (i + 1)])
        assert (kcodes[i], sigs[sg * i: sg * (i + 1)] if w[0] == 0 else None) == (w[0], w[1] if w[0] == 0 else None), i


@pytest.mark.parametrize("n", [1 << 16, (1 << 17) + 37, (1 << 18) + 74])
def test_device_batch_sign_verify_pipeline_2pow16(orc, n):
    """device-resident: 2^16 .. 2^18 keys -> public keys -> deterministic signatures (per-item t) -> verification, one
    stream, no host round trip; a sample is compared with the oracle, all of it must verify.  (The three sizes take the
    hashing kernels of the signing side through workgroups of 256, 512 and 1024 lanes.)"""
    eng = engine()
    l, no, sg = 128, 32, 48
    oid = E.LEVEL_OID[l]
    privs = dev(orc.fill(no * n, 0xD16))
    hashes = dev(orc.fill(no * n, 0xE27))
    ts = dev(orc.fill(8 * n, 0xF38))
    pubs = torch.empty(2 * no * n, dtype=torch.uint8, device="cuda")
    sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
    c1 = torch.empty(n, dtype=torch.int32, device="cuda")
    c2 = torch.empty(n, dtype=torch.int32, device="cuda")
    c3 = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, c1)
    eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c2, t=ts, t_len=8, t_shared=False)
    eng.bignVerifyL_batch_dev(l, oid, hashes, sigs, pubs, c3)
    torch.cuda.synchronize()
    assert int(c1.abs().sum()) == 0 and int(c2.abs().sum()) == 0 and int(c3.abs().sum()) == 0
    hp, hh, ht, hs, hq = host(privs), host(hashes), host(ts), host(sigs), host(pubs)
    for i in list(range(0, n, 997)) + [n - 1]:
        assert orc.sign2(l, oid, hh[no * i: no * (i + 1)], hp[no * i: no * (i + 1)], ht[8 * i: 8 * (i + 1)]) == (0, hs[sg * i: sg * (i + 1)])
        assert orc.pubkey_calc(l, hp[no * i: no * (i + 1)]) == (0, hq[2 * no * i: 2 * no * (i + 1)])
    # same hashes and keys, shared t: a different nonce, still valid
    eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c2, t=ts[:8].clone(), t_len=8, t_shared=True)
    eng.bignVerifyL_batch_dev(l, oid, hashes, sigs, pubs, c3)
    torch.cuda.synchronize()
    assert int(c2.abs().sum()) == 0 and int(c3.abs().sum()) == 0
    assert orc.sign2(l, oid, hh[:no], hp[:no], ht[:8]) == (0, host(sigs)[:sg])


def test_misuse_and_argument_checks():
    """argument checks in the reference's order: parameters, pointers, OID, rng, private key"""
    eng = engine()
    P = _params(eng, 128)
    oid = E.LEVEL_OID[128]
    h, d = bytes(32), (1).to_bytes(32, "little")
    bad = eng.bignParamsStd(E.CURVE_NAME[128])
    bad.p[0] ^= 1
    assert eng.bignSign2(bad, oid, h, d)[0] == E.ERR_BAD_PARAMS
    assert eng.bignPubkeyCalc(bad, d)[0] == E.ERR_BAD_PARAMS
    assert eng.bignSign2(P, b"\x06\x01", h, d)[0] == E.ERR_BAD_OID
    assert eng.bignSign2(P, oid, h, bytes(32))[0] == 504
    assert eng.lib.bignSign(None, None, oid, 11, h, d, None, None) != 0
    assert eng.bignSign(P, oid, h, d, ctypes_null_rng())[0] == 304
    # overlapping hash / sig buffers are refused (memIsDisjoint2, bign_sign.c:163-165)
    import ctypes
    buf = ctypes.create_string_buffer(80)
    code = eng.lib.bignSign2(buf, ctypes.byref(P), oid, ctypes.c_size_t(11), ctypes.byref(buf, 16), d, None, ctypes.c_size_t(0))
    assert code == E.ERR_BAD_INPUT
    # device entry points: alignment and null checks
    t = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert eng.lib.bee2hip_bignPubkeyCalcL_batch_dev(ctypes.c_size_t(128), None, ctypes.c_size_t(1), None, None, None) == E.ERR_BAD_INPUT
    assert eng.lib.bee2hip_bignPubkeyCalcL_batch_dev(ctypes.c_size_t(100), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(1),
                                                     ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()), None) == E.ERR_BAD_PARAMS


def ctypes_null_rng():
    return E.Engine.GEN_I()


def test_sign_from_many_threads(orc):
    """drop-in signing is re-entrant (per-thread staging, per-thread stream scratch)"""
    eng = engine()
    oid = E.LEVEL_OID[128]
    P = _params(eng, 128)
    errs = []

    def work(seed):
        try:
            eng.set_device(0)
            for i in range(6):
                d = orc.fill(32, seed * 100 + i)
                h = orc.fill(32, seed * 100 + 50 + i)
                want = orc.sign2(128, oid, h, d, None)
                got = eng.bignSign2(P, oid, h, d, None)
                if want != got:
                    errs.append((seed, i))
        except Exception as e:          # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(s,)) for s in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs


@pytest.mark.parametrize("n", [1 << 15, (1 << 15) + 77, 3 << 14, (1 << 17) - 5, (1 << 17) + 1, 1 << 18])
def test_lds_lookup_forms_at_their_own_sizes_match_the_scanning_kernel(n):
    """From 2^15 scalars (round 6; 3 * 2^14 before) on the 256-bit curve k G looks its window entries up in LDS (bign_mulbase_lds_kernel: workgroups of 512
    lanes up to 2^17 scalars, of 1024 above); the same batch through the scanning kernel (forced, experiments build) must give the
    same public keys and -- deterministic signatures -- the same octets; a sample of both is checked against the oracle."""
    import orclib
    orc = orclib.load()
    eng = exp_engine()
    l, no, sg = 128, 32, 48
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    privs = torch.empty(no * n, dtype=torch.uint8, device="cuda")
    privs.view(torch.int64).random_(generator=g)
    privs.view(-1, no)[:, no - 1] &= 0x7F
    privs.view(-1, no)[:, 0] |= 1
    hsh = torch.empty(no * n, dtype=torch.uint8, device="cuda")
    hsh.view(torch.int64).random_(generator=g)
    out = {}
    for form in (0, 1):
        eng.lib.bee2hip_internal_tune(10, form)
        pubs = torch.zeros(2 * no * n, dtype=torch.uint8, device="cuda")
        sigs = torch.zeros(sg * n, dtype=torch.uint8, device="cuda")
        sc = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        eng.bignPubkeyCalcL_batch_dev(l, privs, pubs, sc)
        torch.cuda.synchronize()
        assert int(sc.abs().sum()) == 0
        eng.bignSign2L_batch_dev(l, E.LEVEL_OID[l], hsh, privs, sigs, sc)
        torch.cuda.synchronize()
        assert int(sc.abs().sum()) == 0
        out[form] = (host(pubs), host(sigs))
    eng.lib.bee2hip_internal_tune(10, 0)
    assert out[0] == out[1]
    pv, hv = host(privs), host(hsh)
    for i in (0, 1, 63, 64, 511, 512, 1023, 1024, n // 2, n - 1):
        d, h = pv[no * i: no * i + no], hv[no * i: no * i + no]
        assert orc.pubkey_calc(l, d) == (0, out[0][0][2 * no * i: 2 * no * (i + 1)])
        assert orc.sign2(l, E.LEVEL_OID[l], h, d, None) == (0, out[0][1][sg * i: sg * (i + 1)])
