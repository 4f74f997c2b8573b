"""n signatures under ONE public key (bee2hip_bignVerify_onekey_batch / bee2hip_bignVerifyL_onekey_batch_dev; bign_kernels.hip
"one signer"): the same verdict per signature as bignVerify (bign_sign.c:268-361) -- checked against the oracle, against the
general batch entry on the same inputs with the key repeated, and on the reference's own fixtures."""
import numpy as np
import pytest
import torch

from bee2_amd import engine as E
from gpulib import dev, engine, exp_engine, host

pytestmark = pytest.mark.gpu


def _signed_under_one_key(eng, orc, l, n, seed):
    """n valid signatures of random hashes under one private key (the signing kernels are pinned to the oracle elsewhere;
    a sample is checked here again) -> (pubkey bytes, hashes, sigs) as numpy rows"""
    no, sg = l // 4, 3 * l // 8
    oid = E.LEVEL_OID[l]
    priv = bytearray(orc.fill(no, seed))
    priv[no - 1] &= 0x3F
    priv = bytes(priv)
    code, pub = orc.pubkey_calc(l, priv)
    assert code == 0
    hashes = dev(orc.fill(no * n, seed + 1))
    privs = dev(priv * n)
    sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
    c = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(l, oid, hashes, privs, sigs, c)
    torch.cuda.synchronize()
    assert int(c.abs().sum()) == 0
    H = np.frombuffer(host(hashes), dtype=np.uint8).reshape(n, no).copy()
    S = np.frombuffer(host(sigs), dtype=np.uint8).reshape(n, sg).copy()
    for i in (0, n // 2, n - 1):
        assert orc.sign2(l, oid, H[i].tobytes(), priv) == (0, S[i].tobytes())
    return pub, H, S


def _onekey_dev(eng, l, H, S, pub):
    n = H.shape[0]
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_onekey_batch_dev(l, E.LEVEL_OID[l], dev(H.reshape(-1)), dev(S.reshape(-1)), pub, codes)
    torch.cuda.synchronize()
    return codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


def _general_dev(eng, l, H, S, pub):
    n = H.shape[0]
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    K = np.tile(np.frombuffer(pub, dtype=np.uint8), n)
    eng.bignVerifyL_batch_dev(l, E.LEVEL_OID[l], dev(H.reshape(-1)), dev(S.reshape(-1)), dev(K), codes)
    torch.cuda.synchronize()
    return codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


@pytest.mark.parametrize("l,n", [(128, 1), (128, 63), (128, 5000), (192, 1500), (256, 1200)])
def test_onekey_batch_against_the_oracle(orc, l, n):
    """valid signatures, a third of them damaged (hash, s0, s1; s1 >= q; s0 = 0 and s0 = ff..ff: every window of v empty / full):
    every verdict against the oracle's bignVerify restatement"""
    eng = engine()
    no = l // 4
    pub, H, S = _signed_under_one_key(eng, orc, l, n, 0x1C0 + l + n)
    rng = np.random.default_rng(l * 7 + n)
    bad = rng.choice(n, n // 3, replace=False)
    kind = rng.integers(0, 3, bad.size)
    bit = (1 << rng.integers(0, 8, bad.size)).astype(np.uint8)
    sel = bad[kind == 0]; H[sel, rng.integers(0, no, sel.size)] ^= bit[kind == 0]
    sel = bad[kind == 1]; S[sel, rng.integers(0, no // 2, sel.size)] ^= bit[kind == 1]
    sel = bad[kind == 2]; S[sel, no // 2 + rng.integers(0, no, sel.size)] ^= bit[kind == 2]
    if n >= 60:
        S[5, no // 2:] = 0xFF                         # s1 >= q
        S[6, : no // 2] = 0                           # v = 2^l: only the top window
        S[7, : no // 2] = 0xFF
        S[8, : no // 2] = 0; S[8, 3] = 0x80           # one window of v
    got = _onekey_dev(eng, l, H, S, pub)
    want = np.array(orc.verify_batch_l(l, E.LEVEL_OID[l], H.tobytes(), S.tobytes(), pub * n, nthreads=32), dtype=np.int64)
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, (diff[:5], got[diff[:5]], want[diff[:5]])
    assert int((want == 0).sum()) >= n - n // 3 - 4
    if n >= 60:
        assert want[5] == 510 and 510 in want[bad]
    # the general entry with the key repeated says the same
    assert np.array_equal(_general_dev(eng, l, H, S, pub), want)
    # host-pointer form
    code, hc = eng.bignVerify_onekey_batch(H.tobytes(), S.tobytes(), pub, oid_der=E.LEVEL_OID[l], params=eng.bignParamsStd(E.CURVE_NAME[l]))
    assert code == 0 and np.array_equal(np.array(hc, dtype=np.int64), want)


def test_onekey_reference_fixtures_grouped_by_key(golden):
    """the 2048 genuine triples of the reference (64 key pairs) and its edge fixtures, one call per distinct key: the codes of
    the fixtures -- among them keys off the curve and coordinates >= p (the general path with the key repeated)"""
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    n = len(hs) // 32
    cases = [(hs[32 * i: 32 * i + 32], ss[48 * i: 48 * i + 48], ps[64 * i: 64 * i + 64], 0) for i in range(n)]
    cases += [tuple(bytes.fromhex(c[k]) for k in ("hash", "sig", "pubkey")) + (c["code"],) for c in golden.bign_edge
              if len(bytes.fromhex(c["sig"])) == 48 and len(bytes.fromhex(c["pubkey"])) == 64 and len(bytes.fromhex(c["hash"])) == 32]
    groups = {}
    for h, s, p, code in cases:
        groups.setdefault(p, []).append((h, s, code))
    assert len(groups) >= 64
    seen = set()
    for p, items in groups.items():
        H = np.frombuffer(b"".join(x[0] for x in items), dtype=np.uint8).reshape(-1, 32)
        S = np.frombuffer(b"".join(x[1] for x in items), dtype=np.uint8).reshape(-1, 48)
        got = _onekey_dev(eng, 128, H, S, p)
        want = np.array([x[2] for x in items], dtype=np.int64)
        assert np.array_equal(got, want), (p.hex(), got, want)
        seen |= set(want.tolist())
    assert {0, 505, 510} <= seen


def test_onekey_full_size_against_the_general_entry(orc):
    """2^18 + 5 signatures under one key, 1/16 of them damaged: every verdict equal to the general batch entry's (itself compared with
    the oracle entry by entry in test_gpu_bign.py), a sample against the oracle; a second key evicts nothing it should not"""
    eng = engine()
    l, n = 128, (1 << 18) + 5
    pub, H, S = _signed_under_one_key(eng, orc, l, n, 0x51D)
    rng = np.random.default_rng(0x51D)
    bad = rng.choice(n, n // 16, replace=False)
    S[bad, rng.integers(0, 48, bad.size)] ^= 0x04
    got = _onekey_dev(eng, l, H, S, pub)
    want = _general_dev(eng, l, H, S, pub)
    assert np.array_equal(got, want)
    assert int((got == 510).sum()) == bad.size and int((got == 0).sum()) == n - bad.size
    idx = list(range(0, n, 4099))
    o = orc.verify_batch_l(l, E.LEVEL_OID[l], H[idx].tobytes(), S[idx].tobytes(), pub * len(idx), nthreads=8)
    assert np.array_equal(np.array(o, dtype=np.int64), got[idx])
    # 20 other keys in between, then the first one again
    for k in range(20):
        p2, H2, S2 = _signed_under_one_key(eng, orc, l, 40, 0x700 + k)
        assert not _onekey_dev(eng, l, H2, S2, p2).any()
        assert (_onekey_dev(eng, l, H2, S2, pub) == 510).all()          # the wrong key
    assert np.array_equal(_onekey_dev(eng, l, H[:4096], S[:4096], pub), want[:4096])


def test_onekey_argument_checks(orc):
    eng = engine()
    pub, H, S = _signed_under_one_key(eng, orc, 128, 4, 0x99)
    params = eng.bignParamsStd(E.CURVE_NAME[128])
    assert eng.bignVerify_onekey_batch(H.tobytes(), S.tobytes(), pub, oid_der=b"\x06\x01", params=params)[0] == E.ERR_BAD_OID
    assert eng.bignVerify_onekey_batch(b"", b"", pub, params=params) == (0, [])
    bad = bytearray(pub); bad[0] ^= 1                                     # off the curve: the general path's verdicts
    code, c1 = eng.bignVerify_onekey_batch(H.tobytes(), S.tobytes(), bytes(bad), params=params)
    code2, c2 = eng.bignVerify_batch(H.tobytes(), S.tobytes(), bytes(bad) * 4, params=params)
    assert code == 0 and code2 == 0 and c1 == c2 and all(c != 0 for c in c1)
    big = b"\xff" * 64                                                    # coordinates >= p
    assert eng.bignVerify_onekey_batch(H.tobytes(), S.tobytes(), big, params=params) == (0, [505] * 4)


@pytest.mark.parametrize("l", [128, 192, 256])
def test_onekey_both_table_forms(orc, l):
    """the 8-bit table of a key and the 16-bit one it gets after enough signatures, each FORCED (experiments build, tune 20), and
    the switch from one to the other in mid-life of a key: verdicts of the general entry, valid and damaged signatures"""
    eng = exp_engine()
    tune = eng.lib.bee2hip_internal_tune
    n = 3000
    pub, H, S = _signed_under_one_key(eng, orc, l, n, 0x616 + l)
    S[::7, 2] ^= 0x40
    S[5, : l // 8] = 0
    S[6, : l // 8] = 0xFF
    want = _general_dev(eng, l, H, S, pub)
    assert int((want == 510).sum()) >= n // 7 and int((want == 0).sum()) >= n - n // 7 - 4
    try:
        assert tune(20, 63) == 0                       # never: 8-bit windows
        assert np.array_equal(_onekey_dev(eng, l, H, S, pub), want)
        assert tune(20, 13) == 0                       # after 2^13 signatures: this key has 3000 behind it, two more calls cross the line
        assert np.array_equal(_onekey_dev(eng, l, H, S, pub), want)
        assert np.array_equal(_onekey_dev(eng, l, H, S, pub), want)
        assert np.array_equal(_onekey_dev(eng, l, H[:100], S[:100], pub), want[:100])
        assert tune(20, 0) == 0                        # a new key gets both tables at once
        pub2, H2, S2 = _signed_under_one_key(eng, orc, l, 257, 0x717 + l)
        S2[3, 1] ^= 1
        assert np.array_equal(_onekey_dev(eng, l, H2, S2, pub2), _general_dev(eng, l, H2, S2, pub2))
        o = orc.verify_batch_l(l, E.LEVEL_OID[l], H2.tobytes(), S2.tobytes(), pub2 * 257, nthreads=8)
        assert np.array_equal(np.array(o, dtype=np.int64), _onekey_dev(eng, l, H2, S2, pub2))
    finally:
        tune(20, -1)


def _keyed_case(eng, orc, l, nkeys, n, seed, bogus=()):
    """n signatures of nkeys signers, round robin with a twist; bogus = indices of keys replaced by junk AFTER signing"""
    no, sg = l // 4, 3 * l // 8
    oid = E.LEVEL_OID[l]
    privs = []
    for k in range(nkeys):
        d = bytearray(orc.fill(no, seed + 31 * k)); d[no - 1] &= 0x3F
        privs.append(bytes(d))
    pubs = [orc.pubkey_calc(l, d)[1] for d in privs]
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, nkeys, n).astype(np.uint32)
    hashes = dev(orc.fill(no * n, seed + 1))
    dd = dev(b"".join(privs[int(k)] for k in idx))
    sigs = torch.empty(sg * n, dtype=torch.uint8, device="cuda")
    c = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(l, oid, hashes, dd, sigs, c)
    torch.cuda.synchronize()
    assert int(c.abs().sum()) == 0
    H = np.frombuffer(host(hashes), dtype=np.uint8).reshape(n, no).copy()
    S = np.frombuffer(host(sigs), dtype=np.uint8).reshape(n, sg).copy()
    for k in bogus:
        p = bytearray(pubs[k])
        if k % 2:
            p[3] ^= 0x20                                  # off the curve
        else:
            p[:no] = b"\xff" * no                         # x >= p
        pubs[k] = bytes(p)
    return pubs, idx, H, S


@pytest.mark.parametrize("l,nkeys,n", [(128, 1, 300), (128, 64, 6000), (128, 700, 3000), (192, 9, 900), (256, 5, 700)])
def test_keyed_batch_against_the_oracle(orc, l, nkeys, n):
    """n signatures of K signers (bee2hip_bignVerifyL_keyed_batch_dev and the host-pointer form): a fifth damaged, two of the keys junk
    (off the curve: the slow kernel for their signatures; a coordinate >= p: ERR_BAD_PUBKEY), indices out of range -- the oracle's
    bignVerify with the signature's key on every entry, and the general entry with the keys expanded"""
    eng = engine()
    no = l // 4
    bogus = (1, 2) if nkeys >= 5 else ()
    pubs, idx, H, S = _keyed_case(eng, orc, l, nkeys, n, 0xE00 + l + nkeys, bogus)
    rng = np.random.default_rng(nkeys)
    bad = rng.choice(n, n // 5, replace=False)
    S[bad, rng.integers(0, S.shape[1], bad.size)] ^= (1 << rng.integers(0, 8, bad.size)).astype(np.uint8)
    K = np.frombuffer(b"".join(pubs), dtype=np.uint8).reshape(nkeys, 2 * no)
    want = np.array(orc.verify_batch_l(l, E.LEVEL_OID[l], H.tobytes(), S.tobytes(), K[idx].tobytes(), nthreads=32), dtype=np.int64)
    idx2 = idx.copy()
    out_of_range = rng.choice(n, 3, replace=False)
    idx2[out_of_range] = nkeys + np.arange(3, dtype=np.uint32) * 1000
    want2 = want.copy(); want2[out_of_range] = E.ERR_BAD_INPUT if hasattr(E, "ERR_BAD_INPUT") else 109
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_keyed_batch_dev(l, E.LEVEL_OID[l], dev(H.reshape(-1)), dev(S.reshape(-1)), b"".join(pubs),
                                    torch.from_numpy(idx2.astype(np.int32)).cuda(), codes)
    torch.cuda.synchronize()
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    diff = np.nonzero(got != want2)[0]
    assert diff.size == 0, (diff[:5], got[diff[:5]], want2[diff[:5]], idx2[diff[:5]])
    if bogus:
        assert (want[idx == 2] == 505).all() and (want[idx == 1] != 0).all()
    assert int((want == 0).sum()) >= n // 4
    # the general entry with every signature's key beside it
    g = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_batch_dev(l, E.LEVEL_OID[l], dev(H.reshape(-1)), dev(S.reshape(-1)), dev(K[idx].reshape(-1)), g)
    torch.cuda.synchronize()
    assert np.array_equal(g.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, want)
    # host-pointer form
    code, hc = eng.bignVerify_keyed_batch(H.tobytes(), S.tobytes(), b"".join(pubs), [int(x) for x in idx2], oid_der=E.LEVEL_OID[l],
                                          params=eng.bignParamsStd(E.CURVE_NAME[l]))
    assert code == 0 and np.array_equal(np.array(hc, dtype=np.int64), want2)


def test_keyed_reference_fixtures_in_one_call(golden):
    """all 2048 genuine triples of the reference and its 433 edge fixtures in ONE call: their distinct keys as the signers (Q = G, keys
    off the curve and >= p among them), the fixtures' codes"""
    eng = engine()
    hs, ss, ps = golden.bign_base_arrays()
    n0 = len(hs) // 32
    cases = [(hs[32 * i: 32 * i + 32], ss[48 * i: 48 * i + 48], ps[64 * i: 64 * i + 64], 0) for i in range(n0)]
    cases += [tuple(bytes.fromhex(c[k]) for k in ("hash", "sig", "pubkey")) + (c["code"],) for c in golden.bign_edge]
    keys = sorted({c[2] for c in cases})
    pos = {k: i for i, k in enumerate(keys)}
    idx = np.array([pos[c[2]] for c in cases], dtype=np.int32)
    H = b"".join(c[0] for c in cases); S = b"".join(c[1] for c in cases)
    codes = torch.full((len(cases),), -1, dtype=torch.int32, device="cuda")
    eng.bignVerifyL_keyed_batch_dev(128, E.LEVEL_OID[128], dev(H), dev(S), b"".join(keys), torch.from_numpy(idx).cuda(), codes)
    torch.cuda.synchronize()
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    want = np.array([c[3] for c in cases], dtype=np.int64)
    diff = np.nonzero(got != want)[0]
    assert diff.size == 0, (diff[:8], got[diff[:8]], want[diff[:8]])
    assert len(keys) >= 64 and {0, 505, 510} <= set(want.tolist())


def test_onekey_and_keyed_from_six_threads_with_a_tiny_cache(orc):
    """six host threads, each on its own stream-less calls: its own key, a key all of them share, and a keyed batch over all the keys --
    with the table cache cut to THREE keys (experiments build, tune 21), so tables are evicted while other threads still queue kernels
    on them (a table lives until its last user lets go; hipFree waits for the device).  Every verdict as computed beforehand."""
    import threading
    eng = exp_engine()
    tune = eng.lib.bee2hip_internal_tune
    l, n, nthr = 128, 700, 6
    cases = [_signed_under_one_key(eng, orc, l, n, 0xA00 + 13 * t) for t in range(nthr + 1)]      # [nthr] = the shared key
    for pub, H, S in cases:
        S[::9, 7] ^= 2
    want = [_general_dev(eng, l, H, S, pub) for pub, H, S in cases]
    pubs_all = b"".join(c[0] for c in cases)
    Hk = np.concatenate([c[1] for c in cases]); Sk = np.concatenate([c[2] for c in cases])
    idxk = np.repeat(np.arange(nthr + 1, dtype=np.int32), n)
    wantk = np.concatenate(want)
    errors = []
    start = threading.Barrier(nthr)

    def run(t):
        try:
            torch.cuda.set_device(0)
            eng.set_device(0)
            start.wait()
            for r in range(12):
                for j in (t, nthr):
                    pub, H, S = cases[j]
                    if not np.array_equal(_onekey_dev(eng, l, H, S, pub), want[j]):
                        errors.append((t, r, j)); return
                codes = torch.full((idxk.size,), -1, dtype=torch.int32, device="cuda")
                eng.bignVerifyL_keyed_batch_dev(l, E.LEVEL_OID[l], dev(Hk.reshape(-1)), dev(Sk.reshape(-1)), pubs_all,
                                                torch.from_numpy(idxk).cuda(), codes)
                torch.cuda.synchronize()
                if not np.array_equal(codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, wantk):
                    errors.append((t, r, "keyed")); return
        except Exception as e:                                            # noqa: BLE001
            errors.append((t, repr(e)))
    try:
        assert tune(21, 3) == 0
        threads = [threading.Thread(target=run, args=(t,)) for t in range(nthr)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    finally:
        tune(21, 1024)
    assert not errors, errors[:5]


def test_onekey_entry_replays_from_a_hip_graph(orc):
    """once a key has its table (one eager call on the stream) the one-signer _dev entry is plain stream work: captured into a
    hipGraph and replayed on fresh signatures in the same buffers (INTEGRATION "hipGraph capture")"""
    eng = engine()
    l, n = 128, 5000
    pub, H, S = _signed_under_one_key(eng, orc, l, 3 * n, 0xC0DE)
    dh = torch.empty(32 * n, dtype=torch.uint8, device="cuda"); ds = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    cap = torch.cuda.Stream()

    def load(part, damage):
        Sp = S[part * n:(part + 1) * n].copy()
        Sp[damage, 9] ^= 0x08
        with torch.cuda.stream(cap):
            dh.copy_(torch.from_numpy(H[part * n:(part + 1) * n].reshape(-1).copy()).cuda())
            ds.copy_(torch.from_numpy(Sp.reshape(-1)).cuda())
        cap.synchronize()

    load(0, [])
    with torch.cuda.stream(cap):
        eng.bignVerifyL_onekey_batch_dev(l, E.LEVEL_OID[l], dh, ds, pub, codes)      # eager: the key's table, the stream's scratch
    cap.synchronize()
    assert not codes.any()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        eng.bignVerifyL_onekey_batch_dev(l, E.LEVEL_OID[l], dh, ds, pub, codes)
    for part, damage in ((1, [3, 77, 4999]), (2, [0]), (0, list(range(0, n, 500)))):
        load(part, damage)
        codes.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        want = np.zeros(n, dtype=np.int64); want[damage] = 510
        assert np.array_equal(got, want), (part, np.nonzero(got != want)[0][:5])


def test_captured_graph_survives_the_eviction_of_its_key(orc):
    """a stream capture bakes the key's table addresses into the graph, and the table cache drops its least recently used
    keys.  With a cache of TWO keys (tune 21) the captured key is pushed out by eight other signers, device memory of the tables' size
    is allocated and overwritten in between -- the replay must still give the right verdicts: tables a capture referred to are pinned."""
    eng = exp_engine()
    tune = eng.lib.bee2hip_internal_tune
    l, n = 128, 3000
    pub, H, S = _signed_under_one_key(eng, orc, l, 2 * n, 0xCA97)
    dh = torch.empty(32 * n, dtype=torch.uint8, device="cuda"); ds = torch.empty(48 * n, dtype=torch.uint8, device="cuda")
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    cap = torch.cuda.Stream()

    def load(part, damage):
        Sp = S[part * n:(part + 1) * n].copy()
        Sp[damage, 9] ^= 0x08
        with torch.cuda.stream(cap):
            dh.copy_(torch.from_numpy(H[part * n:(part + 1) * n].reshape(-1).copy()).cuda())
            ds.copy_(torch.from_numpy(Sp.reshape(-1)).cuda())
        cap.synchronize()
    try:
        tune(21, 2)
        load(0, [])
        with torch.cuda.stream(cap):
            eng.bignVerifyL_onekey_batch_dev(l, E.LEVEL_OID[l], dh, ds, pub, codes)      # eager: the key's table
        cap.synchronize()
        assert not codes.any()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            eng.bignVerifyL_onekey_batch_dev(l, E.LEVEL_OID[l], dh, ds, pub, codes)
        builds0 = eng.lib.bee2hip_internal_stat(3)
        for k in range(8):                                   # eight other signers: the cache holds two
            pk, Hk, Sk = _signed_under_one_key(eng, orc, l, 64, 0xD000 + 7 * k)
            assert not _onekey_dev(eng, l, Hk, Sk, pk).any()
        assert eng.lib.bee2hip_internal_stat(3) - builds0 == 8
        junk = [torch.full((300 << 10,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(64)]   # what a freed 278 KiB table would be reused for
        torch.cuda.synchronize()
        for part, damage in ((1, [5, 2999]), (0, [0, 17])):
            load(part, damage)
            codes.fill_(-1)
            graph.replay()
            torch.cuda.synchronize()
            got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            want = np.zeros(n, dtype=np.int64); want[damage] = 510
            assert np.array_equal(got, want), (part, np.nonzero(got != want)[0][:5])
        del junk
    finally:
        tune(21, 1024)


@pytest.mark.parametrize("l", [128, 192, 256])
@pytest.mark.parametrize("quads", [1, 0])
def test_onekey_one_and_four_lanes_per_signature(orc, l, quads):
    """the four-lane form (the product's choice up to 2^16 signatures) and the one-lane form, each FORCED (tune 22) over both table forms,
    one signer and a few: sizes around the quad / wavefront / workgroup edges, damaged signatures, s0 with empty and full windows (a
    lane of the quad with nothing to add), junk keys and indices out of range in the keyed batch -- the oracle on every entry"""
    eng = exp_engine()
    tune = eng.lib.bee2hip_internal_tune
    no = l // 4
    try:
        assert tune(22, quads) == 0
        for n, t16 in ((1, 63), (5, 0), (63, 63), (64, 0), (65, 63), (1023, 0), (2100, 63)):
            assert tune(20, t16) == 0
            pub, H, S = _signed_under_one_key(eng, orc, l, n, 0x4A00 + l + n)
            S[::6, 1] ^= 0x80
            if n > 8:
                S[2, : no // 2] = 0                      # only the top window of v: three lanes of the quad have no window of v
                S[3, : no // 2] = 0; S[3, 1] = 7         # one window of v
                S[4, : no // 2] = 0xFF
                H[7] = 0
            want = np.array(orc.verify_batch_l(l, E.LEVEL_OID[l], H.tobytes(), S.tobytes(), pub * n, nthreads=16), dtype=np.int64)
            got = _onekey_dev(eng, l, H, S, pub)
            assert np.array_equal(got, want), (n, t16, np.nonzero(got != want)[0][:5])
        pubs, idx, H, S = _keyed_case(eng, orc, l, 6, 1500, 0x4B00 + l, bogus=(1, 2))
        S[::5, 2] ^= 4
        K = np.frombuffer(b"".join(pubs), dtype=np.uint8).reshape(6, 2 * no)
        want = np.array(orc.verify_batch_l(l, E.LEVEL_OID[l], H.tobytes(), S.tobytes(), K[idx].tobytes(), nthreads=16), dtype=np.int64)
        idx2 = idx.copy(); idx2[[0, 777]] = [6, 1 << 30]; want[[0, 777]] = E.ERR_BAD_INPUT
        codes = torch.full((1500,), -1, dtype=torch.int32, device="cuda")
        eng.bignVerifyL_keyed_batch_dev(l, E.LEVEL_OID[l], dev(H.reshape(-1)), dev(S.reshape(-1)), b"".join(pubs),
                                        torch.from_numpy(idx2.astype(np.int32)).cuda(), codes)
        torch.cuda.synchronize()
        got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert np.array_equal(got, want), np.nonzero(got != want)[0][:5]
    finally:
        tune(22, -1); tune(20, -1)


def test_onekey_batch_of_2pow22_signatures(orc):
    """2^22 + 3 signatures under one key in ONE call (4 GiB of scratch), every 4099th damaged: verdicts by position, a sample against the
    oracle"""
    eng = engine()
    l, n = 128, (1 << 22) + 3
    oid = E.LEVEL_OID[l]
    d = bytearray(orc.fill(32, 0x2222)); d[31] &= 0x3F
    pub = orc.pubkey_calc(l, bytes(d))[1]
    g = torch.Generator(device="cuda"); g.manual_seed(22)
    h = torch.empty(32 * n + 8, dtype=torch.uint8, device="cuda"); h[: (32 * n) // 8 * 8].view(torch.int64).random_(generator=g)
    h = h[: 32 * n]
    sigs = torch.empty(48 * n, dtype=torch.uint8, device="cuda"); c = torch.empty(n, dtype=torch.int32, device="cuda")
    eng.bignSign2L_batch_dev(l, oid, h, dev(bytes(d)).repeat(n), sigs, c)
    torch.cuda.synchronize()
    assert int(c.abs().sum()) == 0
    sigs.view(n, 48)[::4099, 30] ^= 1
    codes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    for _ in range(2):                                  # (the second call: the key has its 16-bit table)
        eng.bignVerifyL_onekey_batch_dev(l, oid, h, sigs, pub, codes)
        torch.cuda.synchronize()
        want = torch.zeros(n, dtype=torch.int32, device="cuda"); want[::4099] = 510
        assert bool((codes == want).all())
    idx = list(range(0, n, 65537))[:40] + [4099 * 7, n - 1]
    hh, ss = host(h.view(n, 32)[idx].reshape(-1)), host(sigs.view(n, 48)[idx].reshape(-1))
    o = orc.verify_batch_l(l, oid, hh, ss, pub * len(idx), nthreads=8)
    assert o == [int(x) for x in codes[idx].cpu().numpy()]


def test_key_tables_are_built_once_per_key(orc):
    """the cache at work (experiments build: bee2hip_internal_stat 3 = key tables built so far): a key met again builds nothing, a keyed
    batch builds one table per NEW distinct key, a key off the curve none"""
    import ctypes
    eng = exp_engine()
    stat = eng.lib.bee2hip_internal_stat
    stat.restype = ctypes.c_ulonglong
    l = 128
    pub, H, S = _signed_under_one_key(eng, orc, l, 50, 0x7A1)
    b0 = stat(3)
    assert not _onekey_dev(eng, l, H, S, pub).any()
    assert stat(3) == b0 + 1
    for _ in range(3):
        assert not _onekey_dev(eng, l, H, S, pub).any()
    assert stat(3) == b0 + 1
    pubs, idx, Hk, Sk = _keyed_case(eng, orc, l, 9, 400, 0x7A2, bogus=(1,))       # key 1: off the curve
    codes = torch.full((400,), -1, dtype=torch.int32, device="cuda")
    for _ in range(2):
        eng.bignVerifyL_keyed_batch_dev(l, E.LEVEL_OID[l], dev(Hk.reshape(-1)), dev(Sk.reshape(-1)), b"".join(pubs + [pub, pubs[0]]),
                                        torch.from_numpy(idx.astype(np.int32)).cuda(), codes)
        torch.cuda.synchronize()
        assert stat(3) == b0 + 1 + 8                                                # 8 new keys on the curve; `pub` cached, pubs[0] twice
    got = codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert (got[idx != 1] == 0).all() and (got[idx == 1] != 0).all()


@pytest.mark.parametrize("l", [128, 192, 256])
@pytest.mark.parametrize("quads", [1, 0])
def test_keyed_batches_with_busy_and_quiet_keys(orc, l, quads):
    """a keyed batch in which SOME signers have their 16-bit table and the others only the 8-bit one (lanes of one wavefront on different
    table forms), then all of them busy; one / four lanes per signature forced; junk keys among them -- the oracle on every entry"""
    eng = exp_engine()
    tune = eng.lib.bee2hip_internal_tune
    no = l // 4
    try:
        assert tune(22, quads) == 0 and tune(20, 63) == 0
        pubs, idx, H, S = _keyed_case(eng, orc, l, 7, 1100, 0x5C00 + l, bogus=(3,))
        S[::4, 5] ^= 0x20
        S[9, : no // 2] = 0; S[10, : no // 2] = 0xFF; S[11, : no // 2] = 0; S[11, 2] = 1
        K = np.frombuffer(b"".join(pubs), dtype=np.uint8).reshape(7, 2 * no)
        want = np.array(orc.verify_batch_l(l, E.LEVEL_OID[l], H.tobytes(), S.tobytes(), K[idx].tobytes(), nthreads=16), dtype=np.int64)

        def keyed():
            codes = torch.full((1100,), -1, dtype=torch.int32, device="cuda")
            eng.bignVerifyL_keyed_batch_dev(l, E.LEVEL_OID[l], dev(H.reshape(-1)), dev(S.reshape(-1)), b"".join(pubs),
                                            torch.from_numpy(idx.astype(np.int32)).cuda(), codes)
            torch.cuda.synchronize()
            return codes.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert np.array_equal(keyed(), want)                           # every key on its 8-bit table
        assert tune(20, 0) == 0
        for k in (0, 4, 5):                                            # three of them become busy (one-signer calls give them the table)
            sel = idx == k
            assert np.array_equal(_onekey_dev(eng, l, H[sel], S[sel], pubs[k]), want[sel])
        assert tune(20, 63) == 0
        assert np.array_equal(keyed(), want)                           # mixed forms in one batch
        assert tune(20, 0) == 0
        assert np.array_equal(keyed(), want)                           # the keyed call itself makes the rest busy
        assert np.array_equal(keyed(), want)
    finally:
        tune(22, -1); tune(20, -1)
