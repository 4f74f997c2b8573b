import pytest
"""CPU tests: the oracle (oracle/liboracle.so) against the committed golden vectors,
i.e. against the STB annex KATs and outputs of the reference itself
(tools/make_golden.py).  Mirrors test/crypto/{bash,belt,bign,bign128}_test.c."""


def test_belt_sbox_matches_standard(orc, golden):
    assert orc.beltH() == golden.H                    # belt_test.c:174-177


def test_bashF_A2(orc, golden):
    k = golden.kat["bashF_A2"]                        # bash_test.c:41-57
    assert orc.bashF(bytes.fromhex(k["in"])).hex() == k["out"]


def test_bashF_random_batch(orc, golden):
    assert orc.bashF_batch(golden.bashf_in) == golden.bashf_out
    assert orc.bashF_batch(golden.bashf_in, nthreads=4) == golden.bashf_out


def test_bash_hash_A3(orc, golden):
    for k in golden.kat["bash_hash"]:                 # bash_test.c:58-154
        code, d = orc.bashHash(k["l"], golden.H[: k["len"]])
        assert code == 0 and d.hex() == k["out"], k["name"]
        n = k["len"]
        assert orc.bashHash_steps(k["l"], golden.H[:n], [n // 3, n - n // 3]).hex() == k["out"]


def test_bash_hash_bad_level(orc):
    assert orc.bashHash(0, b"")[0] == 502             # ERR_BAD_PARAMS, bash_hash.c:122-123
    assert orc.bashHash(24, b"")[0] == 502
    assert orc.bashHash(272, b"")[0] == 502


def test_belt_block_A1(orc, golden):
    k = golden.kat["belt_block_A1"]                   # belt_test.c:178-184
    assert orc.block_encr(bytes.fromhex(k["in"]), bytes.fromhex(k["key"])).hex() == k["out"]


def test_belt_ctr_A15_A16(orc, golden):
    for k in golden.kat["belt_ctr"]:                  # belt_test.c:423-451
        msg, key, iv = (bytes.fromhex(k[x]) for x in ("in", "key", "iv"))
        assert orc.ctr(msg, key, iv, k["splits"]).hex() == k["out"], k["name"]
        assert orc.ctr(msg, key, iv).hex() == k["out"]


def test_belt_mac_A17(orc, golden):
    for k in golden.kat["belt_mac"]:                  # belt_test.c:452-472
        msg, key = bytes.fromhex(k["in"]), bytes.fromhex(k["key"])
        assert orc.mac(msg, key).hex() == k["out"]
        assert orc.mac_steps(msg, key, [len(msg) // 2, len(msg) - len(msg) // 2]).hex() == k["out"]


def test_belt_hash_A23(orc, golden):
    for k in golden.kat["belt_hash"]:                 # belt_test.c:593-627
        assert orc.belt_hash(golden.H[: k["len"]]).hex() == k["out"]


def test_belt_bash_random_cases(orc, golden):
    for c in golden.belt_bash:
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        assert orc.ctr(msg, key, iv, c["splits"]).hex() == c["ctr"]
        assert orc.ctr(msg, key, iv).hex() == c["ctr"]
        assert orc.mac(msg, key).hex() == c["mac"]
        assert orc.mac_steps(msg, key, c["splits"]).hex() == c["mac"]
        assert orc.belt_hash(msg).hex() == c["belt_hash"]
        assert orc.bashHash(128, msg)[1].hex() == c["bash256"]
        assert orc.bashHash(192, msg)[1].hex() == c["bash384"]
        assert orc.bashHash(256, msg)[1].hex() == c["bash512"]
        assert orc.bashHash_steps(256, msg, c["splits"]).hex() == c["bash512"]


def test_config0_bash256_1MiB(orc, golden):
    """BASELINE.json configs[0]: bash256 of a 1 MiB buffer on the CPU path."""
    big = orc.fill(golden.big["len"], golden.big["seed"])
    assert orc.bashHash(128, big)[1].hex() == golden.big["bash256"]
    assert orc.bashHash(256, big)[1].hex() == golden.big["bash512"]
    assert orc.belt_hash(big).hex() == golden.big["belt_hash"]
    assert orc.mac(big, golden.H[128:160]).hex() == golden.big["belt_mac_keyA17"]


def test_bign_G2_G3(orc, golden):
    for k in golden.kat["bign_verify"]:               # bign_test.c:338-357,388-400
        got = orc.verify(*(bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey")))
        assert got == k["code"], k["name"]


def test_bign_edge_cases(orc, golden):
    for k in golden.bign_edge:
        got = orc.verify(*(bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey")))
        assert got == k["code"], k["name"]


def test_bign_base_batch(orc, golden):
    hs, ss, ps = golden.bign_base_arrays()
    n = 512
    codes = orc.verify_batch(hs[: 32 * n], ss[: 48 * n], ps[: 64 * n], nthreads=4)
    assert codes == [0] * n


def test_mixed_batch_matches_single(orc, golden):
    msgs = orc.fill(8 * 4096, 0x4D1C)
    key = golden.H[128:160]
    dig, tag = orc.mixed_batch(msgs, 4096, key, nthreads=2)
    for i in range(8):
        m = msgs[4096 * i: 4096 * (i + 1)]
        assert dig[64 * i: 64 * i + 64] == orc.bashHash(256, m)[1]
        assert tag[8 * i: 8 * i + 8] == orc.mac(m, key)


def test_belt_ecb_cbc_A9_A12(orc, golden):
    """ECB / CBC incl. ciphertext stealing (belt_test.c:288-396)"""
    for k in golden.kat["belt_modes"]:
        msg, key = bytes.fromhex(k["in"]), bytes.fromhex(k["key"])
        decr = k["fn"].endswith("Decr")
        if "ECB" in k["fn"]:
            code, out = orc.ecb(msg, key, decr)
        elif "CBC" in k["fn"]:
            code, out = orc.cbc(msg, key, bytes.fromhex(k["iv"]), decr)
        elif "BDE" in k["fn"]:                               # belt-bde, A.24-1 / A.25-1 (belt_test.c:628-660)
            code, out = orc.bde(msg, key, bytes.fromhex(k["iv"]), decr)
        else:                                                # belt-sde, A.24-2 / A.25-2 (belt_test.c:661-688)
            assert "SDE" in k["fn"]
            code, out = orc.sde(msg, key, bytes.fromhex(k["iv"]), decr)
        assert code == 0 and out.hex() == k["out"], k["name"]


def _dwp_ops_from_kat(k):
    """the step pattern of belt_test.c:473-543 as orclib ops (E/D consume the plaintext/ciphertext that
    is being transformed, I the open data, A the ciphertext)"""
    crit, op = bytes.fromhex(k["crit"]), bytes.fromhex(k["open"])
    out = bytes.fromhex(k["out"])
    ct = out if k["op"] == "wrap" else crit                  # what is authenticated is always the ciphertext
    pos = {"E": 0, "D": 0, "I": 0, "A": 0}
    src = {"E": crit, "D": crit, "I": op, "A": ct}
    ops = []
    for st in k["steps"]:
        if st[0] == "G":
            ops.append(("G",))
        else:
            ops.append((st[0], src[st[0]][pos[st[0]]: pos[st[0]] + st[1]]))
            pos[st[0]] += st[1]
    return ops


@pytest.mark.parametrize("mode", ["DWP", "CHE"])
def test_belt_dwp_che_A19_A20_and_golden(orc, golden, mode):
    """belt-dwp / belt-che (belt_test.c:473-563) -- STB A.19 / A.20 with the reference's
    incremental pattern (tags taken mid-stream), the reference's outputs on short inputs, and on long seeded ones"""
    g = golden.belt_dwp if mode == "DWP" else golden.belt_che
    for k in g["kat"]:
        key, iv = bytes.fromhex(k["key"]), bytes.fromhex(k["iv"])
        out, macs = orc.dwp_steps(key, iv, _dwp_ops_from_kat(k), mode)
        assert out.hex() == k["out"] and macs[-1].hex() == k["mac"], k["name"]
        crit, op = bytes.fromhex(k["crit"]), bytes.fromhex(k["open"])
        if k["op"] == "wrap":
            assert orc.dwp_wrap(crit, op, key, iv, mode) == (0, bytes.fromhex(k["out"]), bytes.fromhex(k["mac"]))
        else:
            assert orc.dwp_unwrap(crit, op, bytes.fromhex(k["mac"]), key, iv, mode) == (0, bytes.fromhex(k["out"]))
    for c in g["short"]:
        key, iv, crit, op = (bytes.fromhex(c[x]) for x in ("key", "iv", "crit", "open"))
        assert orc.dwp_wrap(crit, op, key, iv, mode) == (0, bytes.fromhex(c["out"]), bytes.fromhex(c["mac"]))
        assert orc.dwp_unwrap(bytes.fromhex(c["out"]), op, bytes.fromhex(c["mac"]), key, iv, mode) == (0, crit)
        bad = bytes([int(c["mac"][:2], 16) ^ 0x80]) + bytes.fromhex(c["mac"])[1:]
        assert orc.dwp_unwrap(bytes.fromhex(c["out"]), op, bad, key, iv, mode)[0] == 511        # ERR_BAD_MAC
    for c in g["long"]:
        key, iv = bytes.fromhex(c["key"]), bytes.fromhex(c["iv"])
        crit, op = orc.fill(c["crit_len"], c["crit_seed"]), orc.fill(c["open_len"], c["open_seed"])
        code, out, mac = orc.dwp_wrap(crit, op, key, iv, mode)
        assert code == 0 and mac.hex() == c["mac"] and orc.belt_hash(out).hex() == c["out_belt_hash"], c["crit_len"]
    assert orc.dwp_wrap(b"x", b"y", b"k" * 31, b"i" * 16, mode)[0] == 109


def test_belt_bde_random_cases(orc, golden):
    """belt-bde of the reference on 1..1000 blocks, all key sizes (tools/make_golden.py bde_random)"""
    assert len(golden.belt_bde) >= 18
    for c in golden.belt_bde:
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        assert orc.bde(msg, key, iv) == (0, bytes.fromhex(c["bde_e"])), c["blocks"]
        assert orc.bde(msg, key, iv, True) == (0, bytes.fromhex(c["bde_d"])), c["blocks"]
        assert orc.bde(bytes.fromhex(c["bde_e"]), key, iv, True)[1] == msg       # D(E(x)) = x
    for bad in (b"", b"x" * 15, b"x" * 17):                                      # whole blocks only, >= 1
        assert orc.bde(bad, b"k" * 32, b"i" * 16)[0] == 109
    assert orc.bde(b"x" * 16, b"k" * 31, b"i" * 16)[0] == 109


def test_belt_sde_random_cases(orc, golden):
    """belt-sde of the reference on sectors of 2..256 blocks (tools/make_golden.py sde_random)"""
    assert len(golden.belt_sde) >= 14
    for c in golden.belt_sde:
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        assert orc.sde(msg, key, iv) == (0, bytes.fromhex(c["sde_e"])), c["blocks"]
        assert orc.sde(msg, key, iv, True) == (0, bytes.fromhex(c["sde_d"])), c["blocks"]
        assert orc.sde(bytes.fromhex(c["sde_e"]), key, iv, True)[1] == msg
        assert orc.wbl(orc.wbl(msg, key)[1], key, True)[1] == msg
    for bad in (b"", b"x" * 16, b"x" * 31, b"x" * 33):                           # >= 2 whole blocks (belt_sde.c:79-80)
        assert orc.sde(bad, b"k" * 32, b"i" * 16)[0] == 109


def test_belt_ecb_cbc_random_cases(orc, golden):
    for c in golden.belt_bash:
        if "ecb_e" not in c:
            continue
        msg, key, iv = (bytes.fromhex(c[x]) for x in ("msg", "key", "iv"))
        assert orc.ecb(msg, key)[1].hex() == c["ecb_e"]
        assert orc.ecb(msg, key, True)[1].hex() == c["ecb_d"]
        assert orc.cbc(msg, key, iv)[1].hex() == c["cbc_e"]
        assert orc.cbc(msg, key, iv, True)[1].hex() == c["cbc_d"]
        assert orc.ecb(orc.ecb(msg, key)[1], key, True)[1] == msg            # D(E(x)) = x
        assert orc.cbc(orc.cbc(msg, key, iv)[1], key, iv, True)[1] == msg
    assert orc.ecb(b"x" * 15, b"k" * 32)[0] == 109                           # count < 16: ERR_BAD_INPUT


def test_bign_big_curves_oracle_vs_golden(orc, golden):
    """bign-curve384v1 / 512v1 (bign192Verify / bign256Verify)"""
    from bee2_amd.engine import LEVEL_OID
    for l in (192, 256):
        d = golden.bign_big[str(l)]
        for t in d["base"][:48]:
            assert orc.verify_l(l, LEVEL_OID[l], *(bytes.fromhex(t[x]) for x in ("hash", "sig", "pubkey"))) == 0
        for e in d["edge"]:
            got = orc.verify_l(l, LEVEL_OID[l], *(bytes.fromhex(e[x]) for x in ("hash", "sig", "pubkey")))
            assert got == e["code"], (l, e["name"])


def test_bign_pubkey_val_oracle_vs_golden(orc, golden):
    """bignPubkeyVal (bign_misc.c:319-365) on the reference-generated cases, all three curves"""
    for l in (128, 192, 256):
        cases = golden.bign_pubkey_val[str(l)]
        bad = [(c["name"], c["code"]) for c in cases if orc.pubkey_val(l, bytes.fromhex(c["pubkey"])) != c["code"]]
        assert not bad, (l, bad[:5])
        pubs = b"".join(bytes.fromhex(c["pubkey"]) for c in cases)
        assert orc.pubkey_val_batch(l, pubs) == [c["code"] for c in cases]
    assert orc.pubkey_val(100, bytes(50)) == 502


def test_bign_oid_lengths_oracle_vs_golden(orc, golden):
    """genuine signatures under OIDs of 3..128 DER octets (reference as signer), all three curves"""
    for c in golden.bign_oid_lengths:
        got = orc.verify_l(c["l"], bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["sig"]),
                           bytes.fromhex(c["pubkey"]))
        assert got == c["code"], (c["l"], len(c["oid"]) // 2)


def test_bign_long_oids_oracle_vs_golden(orc, golden):
    """OIDs of 129 .. 4099 DER octets (reference as signer): the checker itself must agree before the GPU path is held to it"""
    for c in golden.bign_oid_long:
        got = orc.verify_l(c["l"], bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["sig"]),
                           bytes.fromhex(c["pubkey"]))
        assert got == c["code"], (c["l"], len(c["oid"]) // 2)
        if c["code"] == 0:
            assert orc.sign2(c["l"], bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["privkey"]))[1] == \
                bytes.fromhex(c["sig"])


def test_sigvfy_pipeline_oracle_vs_golden(orc, golden):
    """hash -> bignPubkeyVal -> bignVerify as `bee2cmd sig vfy` chains them (cmd_sig.c:461-490), reference verdicts"""
    from bee2_amd.engine import LEVEL_OID
    for l in (128, 192, 256):
        for it in golden.sigvfy_pipeline[str(l)]:
            msg, pub, sig = (bytes.fromhex(it[k]) for k in ("msg", "pubkey", "sig"))
            dig = orc.belt_hash(msg) if l == 128 else orc.bashHash(l, msg)[1]
            assert dig == bytes.fromhex(it["digest"])
            assert orc.pubkey_val(l, pub) == it["pubkey_val"]
            assert orc.verify_l(l, LEVEL_OID[l], dig, sig, pub) == it["verify"]


def test_bign_sign_keygen_oracle_vs_golden(orc, golden):
    """SURVEY 8f-4 tail: the oracle's PubkeyCalc / KeypairGen / Sign / Sign2 against the reference's answers
    (tests/golden/bign_sign.json, tools/make_golden_sign.py), incl. the STB 34.101.45 annex G vectors"""
    S = golden.bign_sign
    k = S["stb"]
    oid = bytes.fromhex(k["oid"])
    g1 = k["G1"]
    code, priv, pub, used = orc.keypair_gen(128, bytes.fromhex(g1["rnd"]))
    assert (code, priv.hex(), pub.hex(), used) == (0, g1["priv"], g1["pub"], 1)
    assert orc.pubkey_calc(128, bytes.fromhex(g1["priv"])) == (0, bytes.fromhex(g1["pub"]))
    for name in ("G2", "G3"):
        c = k[name]
        code, sig, _ = orc.sign_rnd(128, oid, bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]), bytes.fromhex(c["rnd"]))
        assert (code, sig.hex()) == (0, c["sig"]), name
    for name in ("G6", "G7"):
        c = k[name]
        t = None if c["t"] is None else bytes.fromhex(c["t"])
        code, sig = orc.sign2(128, oid, bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]), t)
        assert (code, sig.hex()) == (0, c["sig"]), name
    for l in (128, 192, 256):
        L = S[str(l)]
        no = l // 4
        for c in L["pubkey_calc"]:
            code, pub = orc.pubkey_calc(l, bytes.fromhex(c["priv"]))
            assert code == c["code"] and (code or pub.hex() == c["pub"]), c["priv"]
        for c in L["keypair_gen"]:
            code, priv, pub, used = orc.keypair_gen(l, bytes.fromhex(c["rnd"]))
            if not c["defined"]:              # draw in [q, p): outside the reference multiplier's contract
                assert used * no == c["used"]
                continue
            assert code == c["code"], c
            if code == 0:
                assert (priv.hex(), pub.hex(), used * no) == (c["priv"], c["pub"], c["used"])
        for c in L["sign2"]:
            if c["code"] == 301:
                continue                      # OID syntax is the product's check (pinned by oid_der_cases.json)
            t = None if c["t"] is None else bytes.fromhex(c["t"])
            if t is not None and len(t) > 256:
                continue                      # checker limit
            code, sig = orc.sign2(l, bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]), t)
            assert code == c["code"] and (code or sig.hex() == c["sig"]), c
        for c in L["sign"]:
            code, sig, used = orc.sign_rnd(l, bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["priv"]),
                                           bytes.fromhex(c["rnd"]))
            assert code == c["code"], c
            if code == 0:
                assert (sig.hex(), used * no) == (c["sig"], c["used"])


def test_generic_parameter_sets_python_restatement_vs_reference_fixtures(orc):
    """tests/orc_generic.py (pure-Python bignVerify / bignPubkeyVal over arbitrary parameter sets, the checker of the
    general-curve kernels) against tests/golden/bign_generic.json, whose codes the reference itself produced
    (tools/make_golden_generic.py): isomorphic images of the standard curves (a != -3, reference as signer), random
    primes with crafted no-wrap signatures the reference accepts, damaged variants, malformed parameter sets."""
    import json
    import os
    import orc_generic as OG
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bign_generic.json")))
    PP = [OG.Params.from_hex(c) for c in d["curves"]]
    assert {c["kind"] for c in d["curves"]} == {"iso", "rnd"} and {c["l"] for c in d["curves"]} == {128, 192, 256}
    for c in d["cases"]:
        got = OG.verify(PP[c["curve"]], bytes.fromhex(c["oid"]), bytes.fromhex(c["hash"]), bytes.fromhex(c["sig"]),
                        bytes.fromhex(c["pubkey"]), orc.belt_hash)
        assert got == c["code"], (c["curve"], c["name"])
    assert {c["code"] for c in d["cases"]} == {0, 505, 510}
    for c in d["pubkey_val"]:
        assert OG.pubkey_val(PP[c["curve"]], bytes.fromhex(c["pubkey"])) == c["code"]
    for c in d["bad_params"]:
        P = OG.Params(c["l"], *(bytes.fromhex(c[k]) for k in ("p", "a", "b", "q", "yG")))
        h, s, k = (bytes.fromhex(c[x]) for x in ("hash", "sig", "pubkey"))
        assert OG.verify(P, bytes.fromhex(c["oid"]), h, s, k, orc.belt_hash) == c["verify"], c["name"]
        assert OG.pubkey_val(P, k) == c["pubkey_val"], c["name"]
