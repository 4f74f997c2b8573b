#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE itself (oracle/_ref/libbee2ref.so,
compiled from /root/reference by oracle/Makefile).  Build-container only; the
fixtures it writes are data (inputs + expected outputs) and are committed.

    python tools/make_golden.py

STB hex strings below are the standards' annex vectors as held by the reference's
own tests (test/crypto/bash_test.c:41-154, belt_test.c:178-215,423-472,593-627,
bign_test.c:303-357,388-400, bign128_test.c:101-163); the script re-derives each
one with the reference and refuses to write a fixture that disagrees.
"""
import ctypes
import json
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refgen  # noqa: E402

ROOT = refgen.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")
_sz = ctypes.c_size_t
L = refgen.ref()
H = refgen.beltH()

Q_ORDER = int.from_bytes(bytes.fromhex(
    "07663D2699BF5A7EFC4DFB0DD68E5CD9FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF"), "little")
P_FIELD = 2 ** 256 - 189


def splitmix_bytes(n, seed):
    out = bytearray()
    i = 0
    M = (1 << 64) - 1
    while len(out) < n:
        z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out += struct.pack("<Q", z)
        i += 1
    return bytes(out[:n])


# ----------------------------------------------------------------------------- helpers
def r_bashF(state):
    b = ctypes.create_string_buffer(state, 192)
    L.bashF(b, None)
    return b.raw


def r_bashHash(l, msg):
    out = ctypes.create_string_buffer(l // 4)
    assert L.bashHash(out, _sz(l), msg, _sz(len(msg))) == 0
    return out.raw


def r_keyexpand(key):
    k = (ctypes.c_uint32 * 8)()
    L.beltKeyExpand2(k, key, _sz(len(key)))
    return k


def r_block_encr(block, key):
    b = ctypes.create_string_buffer(block, 16)
    L.beltBlockEncr(b, r_keyexpand(key))
    return b.raw


def r_ctr(msg, key, iv, splits=None):
    st = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTR_keep.restype = _sz
    st = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(st, key, _sz(len(key)), iv)
    buf = ctypes.create_string_buffer(msg, len(msg))
    if splits is None:
        splits = [len(msg)]
    off = 0
    for s in splits:
        L.beltCTRStepE(ctypes.byref(buf, off), _sz(s), st)
        off += s
    assert off == len(msg)
    return buf.raw


def r_mode(fn, msg, key, iv=None):
    """beltECBEncr/Decr, beltCBCEncr/Decr of the reference (belt_ecb.c:109-159, belt_cbc.c:143-193)"""
    out = ctypes.create_string_buffer(len(msg))
    if iv is None:
        code = getattr(L, fn)(out, msg, _sz(len(msg)), key, _sz(len(key)))
    else:
        code = getattr(L, fn)(out, msg, _sz(len(msg)), key, _sz(len(key)), iv)
    assert code == 0, (fn, code)
    return out.raw


def r_mac(msg, key):
    out = ctypes.create_string_buffer(8)
    assert L.beltMAC(out, msg, _sz(len(msg)), key, _sz(len(key))) == 0
    return out.raw


def r_belt_hash(msg):
    out = ctypes.create_string_buffer(32)
    assert L.beltHash(out, msg, _sz(len(msg))) == 0
    return out.raw


def check(name, got, want_hex):
    if got.hex().upper() != want_hex.upper():
        raise SystemExit(f"reference disagrees with STB vector {name}: {got.hex()} != {want_hex}")


# ----------------------------------------------------------------------------- STB KATs
def stb_kats():
    kat = {"beltH": H.hex()}
    # STB 34.101.77 A.2 / A.3
    a2 = ("8FE727775EA7F140B95BB6A200CBB28C7F0809C0C0BC68B7DC5AEDC841BD94E4"
          "03630C301FC255DF5B67DB53EF65E376E8A4D797A6172F2271BA48093173D329"
          "C3502AC946767326A2891971392D3F7089959F5D61621238655975E00E2132A0"
          "D5018CEEDB17731CCD88FC50151D37C0D4A3359506AEDC2E6109511E7703AFBB"
          "014642348D8568AA1A5D9868C4C7E6DFA756B1690C7C2608A2DC136F5997AB8F"
          "BB3F4D9F033C87CA6070E117F099C4094972ACD9D976214B7CED8E3F8B6E058E")
    check("77/A.2", r_bashF(H[:192]), a2)
    kat["bashF_A2"] = {"in": H[:192].hex(), "out": a2.lower()}
    bash = [
        ("A.3.1", 128, 0, "114C3DFAE373D9BCBC3602D6386F2D6A2059BA1BF9048DBAA5146A6CB775709D"),
        ("A.3.2", 128, 127, "3D7F4EFA00E9BA33FEED259986567DCF5C6D12D51057A968F14F06CC0F905961"),
        ("A.3.3", 128, 128, "D7F428311254B8B2D00F7F9EEFBD8F3025FA87C4BABD1BDDBE87E35B7AC80DD6"),
        ("A.3.4", 128, 135, "1393FA1B65172F2D18946AEAE576FA1CF54FDD354A0CB2974A997DC4865D3100"),
        ("A.3.5", 192, 95, "64334AF830D33F63E9ACDFA184E32522103FFF5C6860110A2CD369EDBC04387C"
                           "501D8F92F749AE4DE15A8305C353D64D"),
        ("A.3.6", 192, 96, "D06EFBC16FD6C0880CBFC6A4E3D65AB101FA82826934190FAABEBFBFFEDE93B2"
                           "2B85EA72A7FB3147A133A5A8FEBD8320"),
        ("A.3.7", 192, 108, "FF763296571E2377E71A1538070CC0DE88888606F32EEE6B082788D246686B00"
                            "FC05A17405C5517699DA44B7EF5F55AB"),
        ("A.3.8", 256, 63, "2A66C87C189C12E255239406123BDEDBF19955EAF0808B2AD705E249220845E2"
                           "0F4786FB6765D0B5C48984B1B16556EF19EA8192B985E4233D9C09508D6339E7"),
        ("A.3.9", 256, 64, "07ABBF8580E7E5A321E9B940F667AE209E2952CEF557978AE743DB086BAB4885"
                           "B708233C3F5541DF8AAFC3611482FDE498E58B3379A6622DAC2664C9C118A162"),
        ("A.3.10", 256, 127, "526073918F97928E9D15508385F42F03ADE3211A23900A30131F8A1E3E1EE21C"
                             "C09D13CFF6981101235D895746A4643F0AA62B0A7BC98A269E4507A257F0D4EE"),
        ("A.3.11", 256, 192, "8724C7FF8A2A83F22E38CB9763777B96A70ABA3444F214C763D93CD6D19FCFDE"
                             "6C3D3931857C4FF6CCCD49BD99852FE9EAA7495ECCDD96B571E0EDCF47F89768"),
    ]
    kat["bash_hash"] = []
    for name, l, n, want in bash:
        check("77/" + name, r_bashHash(l, H[:n]), want)
        kat["bash_hash"].append({"name": name, "l": l, "len": n, "out": want.lower()})
    # STB 34.101.31 A.1 block, A.15/A.16 CTR, A.17 MAC, A.23 hash
    check("31/A.1", r_block_encr(H[:16], H[128:160]), "69CCA1C93557C9E3D66BC3E0FA88FA6E")
    kat["belt_block_A1"] = {"in": H[:16].hex(), "key": H[128:160].hex(),
                            "out": "69cca1c93557c9e3d66bc3e0fa88fa6e"}
    a15 = "52C9AF96FF50F64435FC43DEF56BD797D5B5B1FF79FB41257AB9CDF6E63E81F8F00341473EAE409833622DE05213773A"
    check("31/A.15", r_ctr(H[:48], H[128:160], H[192:208], [15, 7, 26]), a15)
    a16 = "DF181ED008A20F43DCBBB93650DAD34B389CDEE5826D40E2D4BD80F49A93F5D212F6333166456F169043CC5F"
    check("31/A.16", r_ctr(H[64:108], H[160:192], H[208:224], [11, 5, 28]), a16)
    kat["belt_ctr"] = [
        {"name": "A.15", "in": H[:48].hex(), "key": H[128:160].hex(), "iv": H[192:208].hex(),
         "splits": [15, 7, 26], "out": a15.lower()},
        {"name": "A.16", "in": H[64:108].hex(), "key": H[160:192].hex(), "iv": H[208:224].hex(),
         "splits": [11, 5, 28], "out": a16.lower()},
    ]
    # STB 34.101.31 A.9 / A.10 (ECB) and A.11 / A.12 (CBC), incl. the ciphertext-stealing cases
    # (belt_test.c:288-396)
    modes = [
        ("A.9-1", "beltECBEncr", H[:48], H[128:160], None,
         "69CCA1C93557C9E3D66BC3E0FA88FA6E5F23102EF109710775017F73806DA9DC46FB2ED2CE771F26DCB5E5D1569F9AB0"),
        ("A.9-2", "beltECBEncr", H[:47], H[128:160], None,
         "69CCA1C93557C9E3D66BC3E0FA88FA6E36F00CFED6D1CA1498C12798F4BEB2075F23102EF109710775017F73806DA9"),
        ("A.10-1", "beltECBDecr", H[64:112], H[160:192], None,
         "0DC5300600CAB840B38448E5E993F421E55A239F2AB5C5D5FDB6E81B40938E2A54120CA3E6E19C7AD750FC3531DAEAB7"),
        ("A.10-2", "beltECBDecr", H[64:100], H[160:192], None,
         "0DC5300600CAB840B38448E5E993F4215780A6E2B69EAFBB258726D7B6718523E55A239F"),
        ("A.11-1", "beltCBCEncr", H[:48], H[128:160], H[192:208],
         "10116EFAE6AD58EE14852E11DA1B8A745CF2480E8D03F1C19492E53ED3A70F60657C1EE8C0E0AE5B58388BF8A68E3309"),
        ("A.11-2", "beltCBCEncr", H[:36], H[128:160], H[192:208],
         "10116EFAE6AD58EE14852E11DA1B8A746A9BBADCAF73F968F875DEDC0A44F6B15CF2480E"),
        ("A.12-1", "beltCBCDecr", H[64:112], H[160:192], H[208:224],
         "730894D6158E17CC1600185A8F411CAB0471FF85C83792398D8924EBD57D03DB95B97A9B7907E4B020960455E46176F8"),
        ("A.12-2", "beltCBCDecr", H[64:100], H[160:192], H[208:224],
         "730894D6158E17CC1600185A8F411CABB6AB7AF8541CF85755B8EA27239F08D2166646E4"),
        # belt-bde (belt_test.c:628-660)
        ("A.24-1", "beltBDEEncr", H[:48], H[128:160], H[192:208],
         "E9CAB32D879CC50C10378EB07C10F26307257E2DBE2B854CBC9F38282D59D6A77F952001C5D1244F53210A27C216D4BB"),
        ("A.25-1", "beltBDEDecr", H[64:112], H[160:192], H[208:224],
         "7041BC226352C706D00EA8EF23CFE46AFAE118577D037FACDC36E4ECC1F6574609F236943FB809E1BEE4A1C686C13ACC"),
        # belt-sde (belt_test.c:661-688)
        ("A.24-2", "beltSDEEncr", H[:48], H[128:160], H[192:208],
         "1FCBB01852003D60B66024C508608BAA2C21AF1E884CF31154D3077D4643CF2249EB2F5A68E4BA019D90211A81D690D9"),
        ("A.25-2", "beltSDEDecr", H[64:112], H[160:192], H[208:224],
         "E9FDF3F788657332E6C46FCF5251B8A6D43543A93E3233837DB1571183A6EF4D7FEB5CDF999E1A3F51A5A3381BEB7FA5"),
    ]
    kat["belt_modes"] = []
    for name, fn, msg, key, iv, want in modes:
        check("31/" + name, r_mode(fn, msg, key, iv), want)
        kat["belt_modes"].append({"name": name, "fn": fn, "in": msg.hex(), "key": key.hex(),
                                  "iv": iv.hex() if iv else None, "out": want.lower()})
    check("31/A.17-1", r_mac(H[:13], H[128:160]), "7260DA60138F96C9")
    check("31/A.17-2", r_mac(H[:48], H[128:160]), "2DAB59771B4B16D0")
    kat["belt_mac"] = [
        {"name": "A.17-1", "in": H[:13].hex(), "key": H[128:160].hex(), "out": "7260da60138f96c9"},
        {"name": "A.17-2", "in": H[:48].hex(), "key": H[128:160].hex(), "out": "2dab59771b4b16d0"},
    ]
    hv = [("A.23-1", 13, "ABEF9725D4C5A83597A367D14494CC2542F20F659DDFECC961A3EC550CBA8C75"),
          ("A.23-2", 32, "749E4C3653AECE5E48DB4761227742EB6DBE13F4A80F7BEFF1A9CF8D10EE7786"),
          ("A.23-3", 48, "9D02EE446FB6A29FE5C982D4B13AF9D3E90861BC4CEF27CF306BFB0B174A154A")]
    kat["belt_hash"] = []
    for name, n, want in hv:
        check("31/" + name, r_belt_hash(H[:n]), want)
        kat["belt_hash"].append({"name": name, "len": n, "out": want.lower()})
    # STB 34.101.45 G.1 key pair, G.2 / G.3 signatures (+ the tests' bit-flip negatives)
    priv = bytes.fromhex("1F66B5B84B7339674533F0329C74F21834281FED0732429E0C79235FC273E269")
    pub = bytes.fromhex("BD1A5650179D79E03FCEE49D4C2BD5DDF54CE46D0CF11E4FF87BF7A890857FD0"
                        "7AC6A60361E8C8173491686D461B2826190C2EDA5909054A9AB84D2AB9D99A90")
    pc = ctypes.create_string_buffer(64)
    assert L.bign128PubkeyCalc(pc, priv) == 0
    check("45/G.1", pc.raw, pub.hex())
    sigs = [("G.2", 13, "E36B7F0377AE4C524027C387FADF1B20CE72F1530B71F2B5FD3A8C584FE2E1AE"
                        "D20082E30C8AF65011F4FB54649DFD3D"),
            ("G.3", 48, "47A63C8B9C936E94B5FAB3D9CBD78366290F3210E163EEC8DB4E921E8479D413"
                        "8F112CC23E6DCE65EC5FF21DF4231C28")]
    kat["bign_verify"] = []
    for name, n, sighex in sigs:
        h = r_belt_hash(H[:n])
        sig = bytes.fromhex(sighex)
        assert refgen.verify(h, sig, pub) == 0, name
        s_bad = bytes([sig[0] ^ 1]) + sig[1:]
        p_bad = bytes([pub[0] ^ 1]) + pub[1:]
        for tag, hh, ss, pp in ((name, h, sig, pub), (name + "/sig^1", h, s_bad, pub),
                                (name + "/pub^1", h, sig, p_bad)):
            kat["bign_verify"].append({"name": tag, "hash": hh.hex(), "sig": ss.hex(),
                                       "pubkey": pp.hex(), "code": refgen.verify(hh, ss, pp)})
    assert [k["code"] for k in kat["bign_verify"]] == [0, 510, 510, 0, 510, 510]
    return kat


# ----------------------------------------------------------------------------- random batches
def bashf_random(n=256, seed=0xBA5F):
    inp = splitmix_bytes(192 * n, seed)
    inp = H[:192] + inp[192:]                      # slot 0 = STB A.2 input
    out = b"".join(r_bashF(inp[192 * i:192 * (i + 1)]) for i in range(n))
    return inp, out


def belt_random(seed=0xBE17):
    import random
    rnd = random.Random(seed)
    cases = []
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 100, 255, 256, 257, 1000, 4096]
    for i, n in enumerate(lens):
        klen = (16, 24, 32)[i % 3]
        key = rnd.randbytes(klen)
        iv = rnd.randbytes(16)
        msg = rnd.randbytes(n)
        splits, left = [], n
        while left:
            s = min(left, rnd.choice((1, 3, 7, 15, 16, 17, 32, 100)))
            splits.append(s)
            left -= s
        modes = {}
        if n >= 16:
            modes = {"ecb_e": r_mode("beltECBEncr", msg, key).hex(), "ecb_d": r_mode("beltECBDecr", msg, key).hex(),
                     "cbc_e": r_mode("beltCBCEncr", msg, key, iv).hex(), "cbc_d": r_mode("beltCBCDecr", msg, key, iv).hex()}
        cases.append({**modes, "key": key.hex(), "iv": iv.hex(), "msg": msg.hex(), "splits": splits,
                      "ctr": r_ctr(msg, key, iv, splits).hex(), "mac": r_mac(msg, key).hex(),
                      "belt_hash": r_belt_hash(msg).hex(),
                      "bash256": r_bashHash(128, msg).hex(), "bash384": r_bashHash(192, msg).hex(),
                      "bash512": r_bashHash(256, msg).hex()})
    return cases


def r_aead_wrap(mode, crit, open_, key, iv):
    """beltDWPWrap / beltCHEWrap of the reference (belt_dwp.c:198-232, belt_che.c:243-277) -> (ciphertext, mac)"""
    d, m = ctypes.create_string_buffer(max(len(crit), 1)), ctypes.create_string_buffer(8)
    f = getattr(L, f"belt{mode}Wrap")
    assert f(d, m, crit, _sz(len(crit)), open_, _sz(len(open_)), key, _sz(len(key)), iv) == 0
    return d.raw[: len(crit)], m.raw


AEAD_KAT = {
    # mode: (protect vector, unprotect vector) = (name, crit, open, key, iv, out hex, mac hex, step pattern);
    # the step patterns are the reference's own (belt_test.c:473-563)
    "DWP": (("A.19-1", H[:16], H[16:48], H[128:160], H[192:208], "52C9AF96FF50F64435FC43DEF56BD797", "3B2E0AEB2B91854B",
             [["E", 7], ["E", 9], ["I", 14], ["G"], ["I", 18], ["G"], ["A", 12], ["G"], ["A", 4], ["G"]]),
            ("A.20-1", H[64:80], H[80:112], H[160:192], H[208:224], "DF181ED008A20F43DCBBB93650DAD34B", "6A2C2C94C4150DC0",
             [["I", 32], ["A", 16], ["D", 16], ["G"]])),
    "CHE": (("A.19-2", H[:15], H[16:48], H[128:160], H[192:208], "BF3DAEAF5D18D2BCC30EA62D2E70A4", "548622B844123FF7",
             [["E", 11], ["E", 4], ["I", 14], ["G"], ["I", 18], ["G"], ["A", 12], ["G"], ["A", 3], ["G"]]),
            ("A.20-2", H[64:84], H[80:112], H[160:192], H[208:224], "2BABF43EB37B5398A9068F31A3C758B762F44AA9",
             "7D9D4F59D40D197D", [["I", 32], ["A", 20], ["D", 20], ["G"]])),
}


def aead_cases(mode, seed):
    """belt-dwp / belt-che of the reference.  kat: the STB vectors with the step pattern of belt_test.c.
    short: (crit, open) pairs in full; long: messages given by (length, splitmix seed) with the mac and
    the belt-hash of the ciphertext -- the lengths cross the chunk boundaries of a parallel Horner
    evaluation (partial blocks on both inputs, empty inputs, >= 1 MiB)."""
    import random
    rnd = random.Random(seed)
    (n1, c1, o1, k1, i1, y1, t1, s1), (n2, c2, o2, k2, i2, x2, t2, s2) = AEAD_KAT[mode]
    y, t = r_aead_wrap(mode, c1, o1, k1, i1)
    check(f"31/{n1} Y", y, y1)
    check(f"31/{n1} T", t, t1)
    xb = ctypes.create_string_buffer(len(c2))
    assert getattr(L, f"belt{mode}Unwrap")(xb, c2, _sz(len(c2)), o2, _sz(len(o2)), bytes.fromhex(t2), k2, _sz(32), i2) == 0
    check(f"31/{n2} X", xb.raw, x2)
    kat = [{"name": n1, "op": "wrap", "crit": c1.hex(), "open": o1.hex(), "key": k1.hex(), "iv": i1.hex(),
            "out": y.hex(), "mac": t.hex(), "steps": s1},
           {"name": n2, "op": "unwrap", "crit": c2.hex(), "open": o2.hex(), "key": k2.hex(), "iv": i2.hex(),
            "out": xb.raw.hex(), "mac": t2.lower(), "steps": s2}]
    short = []
    for i, (nc, no) in enumerate([(0, 0), (0, 1), (1, 0), (15, 15), (16, 16), (17, 17), (16, 0), (0, 16), (31, 33),
                                  (32, 48), (33, 47), (100, 7), (255, 256), (256, 255), (1000, 3), (4096, 4096),
                                  (4097, 1), (1023, 1025)]):
        key, iv = rnd.randbytes((16, 24, 32)[i % 3]), rnd.randbytes(16)
        crit, op = rnd.randbytes(nc), rnd.randbytes(no)
        c, m = r_aead_wrap(mode, crit, op, key, iv)
        short.append({"key": key.hex(), "iv": iv.hex(), "crit": crit.hex(), "open": op.hex(), "out": c.hex(), "mac": m.hex()})
    long_ = []
    for i, (nc, no) in enumerate([(1 << 16, 0), ((1 << 16) + 5, 1000), (16 * 8191, 16 * 4097), ((1 << 20) + 13, 3),
                                  (3, (1 << 20) - 7), (1 << 21, 1 << 18)]):
        key, iv = rnd.randbytes((32, 16, 24)[i % 3]), rnd.randbytes(16)
        crit, op = splitmix_bytes(nc, 0xC417 + i), splitmix_bytes(no, 0x09E7 + i)
        c, m = r_aead_wrap(mode, crit, op, key, iv)
        long_.append({"key": key.hex(), "iv": iv.hex(), "crit_len": nc, "crit_seed": 0xC417 + i, "open_len": no,
                      "open_seed": 0x09E7 + i, "gen": "splitmix64 LE words, x_i = mix(seed+(i+1)*golden)",
                      "out_belt_hash": r_belt_hash(c).hex(), "mac": m.hex()})
    return {"kat": kat, "short": short, "long": long_}


def bde_random(seed=0xBDE):
    """beltBDEEncr / beltBDEDecr of the reference (belt_bde.c:87-133) on whole-block messages;
    block counts straddle the 64-lane wavefront and the per-wavefront chunk boundaries of the kernel"""
    import random
    rnd = random.Random(seed)
    cases = []
    for i, nb in enumerate((1, 2, 3, 4, 5, 31, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 511, 1000)):
        key, iv, msg = rnd.randbytes((16, 24, 32)[i % 3]), rnd.randbytes(16), rnd.randbytes(16 * nb)
        cases.append({"blocks": nb, "key": key.hex(), "iv": iv.hex(), "msg": msg.hex(),
                      "bde_e": r_mode("beltBDEEncr", msg, key, iv).hex(),
                      "bde_d": r_mode("beltBDEDecr", msg, key, iv).hex()})
    return cases


def sde_random(seed=0x5DE):
    """beltSDEEncr / beltSDEDecr of the reference (belt_sde.c:73-121) on sectors of 2..256 blocks: below and
    above the sizes where the reference switches from its base to its rolling-sum form (belt_wbl.c:196-210)"""
    import random
    rnd = random.Random(seed)
    cases = []
    for i, nb in enumerate((2, 3, 4, 5, 6, 7, 8, 16, 31, 32, 33, 64, 100, 256)):
        key, iv, msg = rnd.randbytes((16, 24, 32)[i % 3]), rnd.randbytes(16), rnd.randbytes(16 * nb)
        cases.append({"blocks": nb, "key": key.hex(), "iv": iv.hex(), "msg": msg.hex(),
                      "sde_e": r_mode("beltSDEEncr", msg, key, iv).hex(),
                      "sde_d": r_mode("beltSDEDecr", msg, key, iv).hex()})
    return cases


def int_le(x, n):
    return x.to_bytes(n, "little")


OID_DER = bytes.fromhex("06092A7000020022651F51")
G_Y = int.from_bytes(bytes.fromhex(
    "936A510418CF291E52F608C4663991785D83D651A3C9E45C9FD616FB3CFCF76B"), "little")


def py_mul_base_x(k):
    """x(kG) on bign-curve256v1 with affine big-int arithmetic (fixture signer only)."""
    p = P_FIELD

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1 - 3) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    R, B = None, (0, G_Y)
    while k:
        if k & 1:
            R = add(R, B)
        B = add(B, B)
        k >>= 1
    return R[0]


def bign_sets(seed=0xB164, n_base=2048):
    """base: n_base genuine triples.  edge: crafted cases that hit the exceptional
    branches (Q = +-G, small multiples, R = O, s1 >= q, coordinate >= p, off-curve Q,
    hash >= q), each with the code the reference returns."""
    import random
    rnd = random.Random(seed)
    base = refgen.make_triples(n_base, seed & 0xFFFFFFFF)
    codes = [refgen.verify(*t) for t in base]
    assert all(c == 0 for c in codes)
    edge = []

    def add(name, h, s, p):
        edge.append({"name": name, "hash": h.hex(), "sig": s.hex(), "pubkey": p.hex(),
                     "code": refgen.verify(h, s, p)})

    # small / extreme private keys: Q = dG makes the two NAF tables collide inside ecAddMulA
    for d in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 33, Q_ORDER - 1, Q_ORDER - 2, Q_ORDER - 3,
              Q_ORDER - 16, (Q_ORDER + 1) // 2, 2 ** 128, 2 ** 128 + 1, 2 ** 129 - 1):
        priv = int_le(d, 32)
        pub = ctypes.create_string_buffer(64)
        assert L.bign128PubkeyCalc(pub, priv) == 0
        for k in range(3):
            h = rnd.randbytes(32)
            sig = refgen.sign2(h, priv)
            add(f"d={d if d < 2**64 else hex(d)}/valid{k}", h, sig, pub.raw)
            bad = bytearray(sig)
            bad[rnd.randrange(48)] ^= 1 << rnd.randrange(8)
            add(f"d={d if d < 2**64 else hex(d)}/flip{k}", h, bytes(bad), pub.raw)
        # R = O: pick s0, force s1 = -(s0 + 2^128) d - H  (mod q)
        for k in range(2):
            h = rnd.randbytes(32)
            s0 = rnd.getrandbits(128)
            hh = int.from_bytes(h, "little")
            if hh >= Q_ORDER:
                hh -= Q_ORDER
            s1 = (-(s0 + 2 ** 128) * d - hh) % Q_ORDER
            add(f"d={d if d < 2**64 else hex(d)}/R=O{k}", h, int_le(s0, 16) + int_le(s1, 32), pub.raw)
    # range checks
    t = base[0]
    add("s1=q", t[0], t[1][:16] + int_le(Q_ORDER, 32), t[2])
    add("s1=q-1", t[0], t[1][:16] + int_le(Q_ORDER - 1, 32), t[2])
    add("s1=2^256-1", t[0], t[1][:16] + b"\xff" * 32, t[2])
    add("s1=0", t[0], t[1][:16] + bytes(32), t[2])
    add("s0=0,s1=0", t[0], bytes(48), t[2])
    add("xQ=p", t[0], t[1], int_le(P_FIELD, 32) + t[2][32:])
    add("yQ=p", t[0], t[1], t[2][:32] + int_le(P_FIELD, 32))
    add("xQ=p-1", t[0], t[1], int_le(P_FIELD - 1, 32) + t[2][32:])
    add("xQ=2^256-1", t[0], t[1], b"\xff" * 32 + t[2][32:])
    add("Q=(0,0)", t[0], t[1], bytes(64))
    add("Q=(0,1)", t[0], t[1], bytes(32) + int_le(1, 32))
    add("Q=(1,0)", t[0], t[1], int_le(1, 32) + bytes(32))
    # hash >= q still verifies (one conditional subtraction, bign_sign.c:320-327).  The
    # reference SIGNER requires H < q (zzSubMod precondition), so these are signed here with
    # big-int arithmetic (STB 34.101.45 7.1.3) and only VERIFIED by the reference.
    priv, pub = refgen.keypair(refgen.Combo(99))
    d = int.from_bytes(priv, "little")
    for hv in (Q_ORDER, Q_ORDER + 5, 2 ** 256 - 1, Q_ORDER - 1, 0):
        h = int_le(hv, 32)
        k = rnd.randrange(1, Q_ORDER)
        rx = py_mul_base_x(k)
        s0 = int.from_bytes(r_belt_hash(OID_DER + int_le(rx, 32) + h)[:16], "little")
        s1 = (k - (s0 + 2 ** 128) * d - hv) % Q_ORDER
        add(f"H={hex(hv)[:12]}", h, int_le(s0, 16) + int_le(s1, 32), pub)
    # seeded bit-flip negatives over the base set (the bench recipe)
    for i in range(0, 256):
        h, s, p = (bytearray(x) for x in base[i])
        kind = i % 4
        tgt = (s, s, h, p)[kind]
        lo, hi = ((0, 16), (16, 48), (0, 32), (0, 64))[kind]
        tgt[rnd.randrange(lo, hi)] ^= 1 << rnd.randrange(8)
        add(f"flip{i}/k{kind}", bytes(h), bytes(s), bytes(p))
    return base, edge


def bign_big_curves(seed=0xB192):
    """the 384- and 512-bit curves.  Per level: genuine triples + edge cases
    (small / extreme private keys, R = O, range checks, H >= q, bit flips) with the reference's code."""
    import random
    out = {}
    orders = {192: int.from_bytes(bytes(refparams(192).q)[:48], "little"),
              256: int.from_bytes(bytes(refparams(256).q)[:64], "little")}
    primes = {192: 2 ** 384 - 317, 256: 2 ** 512 - 569}
    for l in (192, 256):
        rnd = random.Random(seed + l)
        no = l // 4
        q, p = orders[l], primes[l]
        base = refgen.make_triples_l(l, 192, seed + l)
        assert all(refgen.verify_l(l, *t) == 0 for t in base)
        edge = []

        def add(name, h, s, pk):
            edge.append({"name": name, "hash": h.hex(), "sig": s.hex(), "pubkey": pk.hex(),
                         "code": refgen.verify_l(l, h, s, pk)})

        for d in (1, 2, 3, 7, 8, 16, q - 1, q - 2, (q + 1) // 2, 2 ** l, 2 ** (l + 1) - 1):
            priv = int_le(d, no)
            pub = refgen.pubkey_calc_l(l, priv)
            h = rnd.randbytes(no)
            sig = refgen.sign2_l(l, h, priv)
            add(f"d={d if d < 2**64 else hex(d)[:14]}/valid", h, sig, pub)
            bad = bytearray(sig)
            bad[rnd.randrange(len(sig))] ^= 1 << rnd.randrange(8)
            add(f"d={d if d < 2**64 else hex(d)[:14]}/flip", h, bytes(bad), pub)
            s0 = rnd.getrandbits(l)
            hh = int.from_bytes(h, "little")
            if hh >= q:
                hh -= q
            s1 = (-(s0 + 2 ** l) * d - hh) % q
            add(f"d={d if d < 2**64 else hex(d)[:14]}/R=O", h, int_le(s0, no // 2) + int_le(s1, no), pub)
        t = base[0]
        half = no // 2
        add("s1=q", t[0], t[1][:half] + int_le(q, no), t[2])
        add("s1=q-1", t[0], t[1][:half] + int_le(q - 1, no), t[2])
        add("s1=max", t[0], t[1][:half] + b"\xff" * no, t[2])
        add("sig=0", t[0], bytes(no + half), t[2])
        add("xQ=p", t[0], t[1], int_le(p, no) + t[2][no:])
        add("yQ=p", t[0], t[1], t[2][:no] + int_le(p, no))
        add("xQ=p-1", t[0], t[1], int_le(p - 1, no) + t[2][no:])
        add("Q=(0,0)", t[0], t[1], bytes(2 * no))
        add("Q=(1,0)", t[0], t[1], int_le(1, no) + bytes(no))
        for i in range(48):
            h, s, pk = (bytearray(x) for x in base[i])
            tgt = (s, s, h, pk)[i % 4]
            tgt[rnd.randrange(len(tgt))] ^= 1 << rnd.randrange(8)
            add(f"flip{i}", bytes(h), bytes(s), bytes(pk))
        out[str(l)] = {"base": [{"hash": h.hex(), "sig": s.hex(), "pubkey": k.hex()} for h, s, k in base],
                       "edge": edge}
    return out


def pubkey_val_cases(seed=0x9B7A):
    """bignPubkeyVal (bign_misc.c:319-365; bign128_test.c:112, bign_test.c:318): per level genuine keys,
    their negatives, G, range violations, bit flips and random pairs, with the reference's return code."""
    import random
    out = {}
    L = refgen.ref()
    for l in (128, 192, 256):
        rnd = random.Random(seed + l)
        no = l // 4
        p = 2 ** (8 * no) - {128: 189, 192: 317, 256: 569}[l]
        prm = refparams(l)
        q = int.from_bytes(bytes(prm.q)[:no], "little")
        yG = bytes(prm.yG)[:no]
        fn = getattr(L, f"bign{l}PubkeyVal")
        cases = []

        def add(name, pk):
            assert len(pk) == 2 * no
            cases.append({"name": name, "pubkey": pk.hex(), "code": fn(pk)})

        keys = []
        for d in [1, 2, 3, q - 1, q - 2, 2 ** l] + [rnd.randrange(1, q) for _ in range(40)]:
            pub = refgen.pubkey_calc_l(l, int_le(d, no))
            keys.append(pub)
            add("valid", pub)
            y = int.from_bytes(pub[no:], "little")
            add("negated", pub[:no] + int_le((p - y) % p, no))
        add("G", bytes(no) + yG)
        add("-G", bytes(no) + int_le(p - int.from_bytes(yG, "little"), no))
        add("(0,0)", bytes(2 * no))
        add("(1,0)", int_le(1, no) + bytes(no))
        k = keys[7]
        add("x=p", int_le(p, no) + k[no:])
        add("y=p", k[:no] + int_le(p, no))
        add("x=p-1", int_le(p - 1, no) + k[no:])
        add("x+p wraps", int_le((int.from_bytes(k[:no], "little") + p) % 2 ** (8 * no), no) + k[no:])
        add("x=max", b"\xff" * no + k[no:])
        add("y=max", k[:no] + b"\xff" * no)
        add("G with y+p", bytes(no) + int_le((int.from_bytes(yG, "little") + p) % 2 ** (8 * no), no))
        for i in range(64):
            pk = bytearray(keys[i % len(keys)])
            pk[rnd.randrange(2 * no)] ^= 1 << rnd.randrange(8)
            add(f"flip{i}", bytes(pk))
        for i in range(32):
            add(f"random{i}", rnd.randbytes(2 * no))
        assert cases[0]["code"] == 0 and {c["code"] for c in cases} == {0, 505}
        out[str(l)] = cases
    return out


def oid_len_cases(seed=0x01DA):
    """Genuine signatures under pre-hash OIDs of many DER lengths (every alignment mod 4, up to 128 octets,
    the kernel's staging limit): bignSign2 / bignVerify of the reference with an explicit oid_der
    (bign_sign.c:140-245, 349-361).  The message hashed at the end is oid || <x_R> || H, so the OID length
    moves every later byte."""
    import random
    L = refgen.ref()
    out = []
    rnd = random.Random(seed)

    def der(body_len):
        body = bytes([0x2A]) + bytes(rnd.randrange(1, 128) for _ in range(body_len - 1))
        return bytes([0x06, body_len]) + body if body_len < 128 else bytes([0x06, 0x81, body_len]) + body

    for l in (128, 192, 256):
        prm = refparams(l)
        no = l // 4
        for body_len in list(range(1, 24)) + [37, 62, 63, 64, 65, 100, 124, 125, 126]:
            oid = der(body_len)
            if len(oid) > 128:
                continue
            priv = int_le(rnd.randrange(1, 2 ** (l - 1)), no)
            pub = refgen.pubkey_calc_l(l, priv)
            h = rnd.randbytes(no)
            sig = ctypes.create_string_buffer(no + no // 2)
            assert L.bignSign2(sig, ctypes.byref(prm), oid, _sz(len(oid)), h, priv, None, _sz(0)) == 0
            for kind in ("valid", "flip"):
                sg = bytearray(sig.raw)
                if kind == "flip":
                    sg[rnd.randrange(len(sg))] ^= 1 << rnd.randrange(8)
                code = L.bignVerify(ctypes.byref(prm), oid, _sz(len(oid)), h, bytes(sg), pub)
                assert code == (0 if kind == "valid" else 510)
                out.append({"l": l, "oid": oid.hex(), "hash": h.hex(), "sig": bytes(sg).hex(), "pubkey": pub.hex(),
                            "code": code})
    return out


def oid_long_cases(seed=0x01DB):
    """The same beyond the 128 octets a kernel stages itself (round 3: the OID's whole 32-byte blocks are pre-hashed once per
    batch, bee2_amd/csrc/bign_kernels.hip make_oid_arg): DER lengths around the block boundaries of the prefix (129 .. 164),
    the one- / two-byte length forms (255 / 256 / 257 octets) and a few long ones.  The private key is kept so that the
    deterministic signature can be reproduced as well."""
    import random
    L = refgen.ref()
    out = []
    rnd = random.Random(seed)

    def der_total(total):
        for hdr, enc in ((2, lambda n: bytes([0x06, n])), (3, lambda n: bytes([0x06, 0x81, n])),
                         (4, lambda n: bytes([0x06, 0x82, n >> 8, n & 255]))):
            n = total - hdr
            if (hdr == 2 and n < 128) or (hdr == 3 and 128 <= n < 256) or (hdr == 4 and 256 <= n < 65536):
                return enc(n) + bytes([0x2A]) + bytes(rnd.randrange(1, 128) for _ in range(n - 1))
        raise ValueError(total)

    for l in (128, 192, 256):
        prm = refparams(l)
        no = l // 4
        for total in (129, 131, 132, 133, 159, 160, 161, 162, 163, 164, 191, 192, 193, 255, 256, 257, 258, 260, 261, 300, 1000, 4099):     # 130 and 259 octets do not exist in DER
            oid = der_total(total)
            assert len(oid) == total
            priv = int_le(rnd.randrange(1, 2 ** (l - 1)), no)
            pub = refgen.pubkey_calc_l(l, priv)
            h = rnd.randbytes(no)
            sig = ctypes.create_string_buffer(no + no // 2)
            assert L.bignSign2(sig, ctypes.byref(prm), oid, _sz(len(oid)), h, priv, None, _sz(0)) == 0
            for kind in ("valid", "flip"):
                sg = bytearray(sig.raw)
                if kind == "flip":
                    sg[rnd.randrange(len(sg))] ^= 1 << rnd.randrange(8)
                code = L.bignVerify(ctypes.byref(prm), oid, _sz(len(oid)), h, bytes(sg), pub)
                assert code == (0 if kind == "valid" else 510)
                out.append({"l": l, "oid": oid.hex(), "hash": h.hex(), "sig": bytes(sg).hex(), "pubkey": pub.hex(),
                            "privkey": priv.hex(), "code": code})
    return out


def sigvfy_pipeline(seed=0x51F7):
    """`bee2cmd sig vfy`-shaped batch (cmd/core/cmd_sig.c:461-490): messages of ragged lengths, their belt-hash
    (level 128) or bash384 / bash512 digest as pre-hash, a public key and a signature per message (reference as
    hasher and signer), plus a few damaged entries with the reference's verdicts."""
    import random
    L = refgen.ref()
    out = {}
    for l in (128, 192, 256):
        rnd = random.Random(seed + l)
        no = l // 4
        items = []
        for i in range(72):
            ln = rnd.choice((0, 1, 31, 32, 33, 64, 95, 127, 128, 129, 200)) if i < 22 else rnd.randrange(0, 1200)
            msg = rnd.randbytes(ln)
            dig = ctypes.create_string_buffer(no)
            if l == 128:
                assert L.beltHash(dig, msg, _sz(ln)) == 0
            else:
                assert L.bashHash(dig, _sz(l), msg, _sz(ln)) == 0
            priv = int_le(rnd.randrange(1, 2 ** (l - 1)), no)
            pub = bytearray(refgen.pubkey_calc_l(l, priv))
            sig = bytearray(refgen.sign2_l(l, dig.raw, priv))
            kind = i % 8
            if kind == 5:
                sig[rnd.randrange(len(sig))] ^= 1 << rnd.randrange(8)
            elif kind == 6:
                pub[rnd.randrange(len(pub))] ^= 1 << rnd.randrange(8)
            elif kind == 7 and ln:
                msg = bytearray(msg); msg[rnd.randrange(ln)] ^= 1; msg = bytes(msg)
                if l == 128:
                    L.beltHash(dig, msg, _sz(ln))
                else:
                    L.bashHash(dig, _sz(l), msg, _sz(ln))
            items.append({"msg": bytes(msg).hex(), "digest": dig.raw.hex(), "sig": bytes(sig).hex(),
                          "pubkey": bytes(pub).hex(),
                          "pubkey_val": getattr(L, f"bign{l}PubkeyVal")(bytes(pub)),
                          "verify": refgen.verify_l(l, dig.raw, bytes(sig), bytes(pub))})
        assert {it["verify"] for it in items} >= {0, 510} and {it["pubkey_val"] for it in items} == {0, 505}
        out[str(l)] = items
    return out


def refparams(l):
    class Params(ctypes.Structure):
        _fields_ = [("l", _sz), ("p", ctypes.c_ubyte * 64), ("a", ctypes.c_ubyte * 64), ("b", ctypes.c_ubyte * 64),
                    ("q", ctypes.c_ubyte * 64), ("yG", ctypes.c_ubyte * 64), ("seed", ctypes.c_ubyte * 8)]
    p = Params()
    name = {128: b"1.2.112.0.2.0.34.101.45.3.1", 192: b"1.2.112.0.2.0.34.101.45.3.2",
            256: b"1.2.112.0.2.0.34.101.45.3.3"}[l]
    assert L.bignParamsStd(ctypes.byref(p), name) == 0
    return p


def oid_cases(seed=0x01D):
    """DER strings with the verdict of the reference's oidFromDER (src/core/oid.c:94-101), which is
    what bignVerify applies to oid_der (bign_sign.c:289-290)."""
    import random
    rnd = random.Random(seed)
    L.oidFromDER.restype = _sz
    cases = []

    def add(der):
        ok = L.oidFromDER(None, bytes(der), _sz(len(der))) != (1 << 64) - 1
        cases.append({"der": bytes(der).hex(), "valid": bool(ok)})

    base = bytes.fromhex("06092A7000020022651F51")
    add(base)
    for i in range(len(base)):                 # single-byte mutations of the belt-hash OID
        for v in (0x00, 0x80, 0xFF, base[i] ^ 0x80, (base[i] + 1) & 0xFF):
            m = bytearray(base)
            m[i] = v
            add(m)
    for cut in range(0, len(base)):
        add(base[:cut])
    add(base + b"\x00")
    for _ in range(300):                       # random well-formed-ish and random garbage
        n = rnd.randrange(0, 20)
        body = bytes(rnd.choice((rnd.randrange(256), rnd.randrange(128), 0x80, 0x81)) for _ in range(n))
        add(bytes([0x06, n]) + body)
        add(bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 12))))
    add(bytes([0x06, 0x81, 0x80]) + bytes([1] * 128))      # long-form length, 128-byte body
    add(bytes([0x06, 0x81, 0x7F]) + bytes([1] * 127))      # non-minimal long form
    add(bytes([0x06, 0x7F]) + bytes([1] * 127))
    return cases


def main():
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "oid_der_cases.json"), "w") as f:
        json.dump(oid_cases(), f)
    with open(os.path.join(GOLD, "bign_big_curves.json"), "w") as f:
        big = bign_big_curves()
        json.dump(big, f)
        from collections import Counter
        for l, d in big.items():
            print(f"bign l={l}: {len(d['base'])} base, edge codes {dict(Counter(e['code'] for e in d['edge']))}")
    with open(os.path.join(GOLD, "bign_pubkey_val.json"), "w") as f:
        json.dump(pubkey_val_cases(), f)
    with open(os.path.join(GOLD, "bign_oid_lengths.json"), "w") as f:
        json.dump(oid_len_cases(), f)
    with open(os.path.join(GOLD, "bign_oid_long.json"), "w") as f:
        json.dump(oid_long_cases(), f)
    with open(os.path.join(GOLD, "sigvfy_pipeline.json"), "w") as f:
        json.dump(sigvfy_pipeline(), f)
    with open(os.path.join(GOLD, "stb_kat.json"), "w") as f:
        json.dump(stb_kats(), f, indent=1)
    inp, out = bashf_random()
    with open(os.path.join(GOLD, "bashf_random.bin"), "wb") as f:
        f.write(inp + out)
    with open(os.path.join(GOLD, "belt_bash_random.json"), "w") as f:
        json.dump(belt_random(), f, indent=1)
    with open(os.path.join(GOLD, "belt_bde_random.json"), "w") as f:
        json.dump(bde_random(), f, indent=1)
    with open(os.path.join(GOLD, "belt_sde_random.json"), "w") as f:
        json.dump(sde_random(), f, indent=1)
    with open(os.path.join(GOLD, "belt_dwp.json"), "w") as f:
        json.dump(aead_cases("DWP", 0xD3B), f, indent=1)
    with open(os.path.join(GOLD, "belt_che.json"), "w") as f:
        json.dump(aead_cases("CHE", 0xC4E), f, indent=1)
    base, edge = bign_sets()
    with open(os.path.join(GOLD, "bign_base.bin"), "wb") as f:      # n x (hash32 | sig48 | pub64)
        for h, s, p in base:
            f.write(h + s + p)
    with open(os.path.join(GOLD, "bign_edge.json"), "w") as f:
        json.dump(edge, f, indent=1)
    # H0: bash256 of 1 MiB (config[0])
    big = splitmix_bytes(1 << 20, 0xBA5F)
    with open(os.path.join(GOLD, "bash256_1MiB.json"), "w") as f:
        json.dump({"seed": 0xBA5F, "len": 1 << 20, "gen": "splitmix64 LE words, x_i = mix(seed+(i+1)*golden)",
                   "bash256": r_bashHash(128, big).hex(), "bash512": r_bashHash(256, big).hex(),
                   "belt_hash": r_belt_hash(big).hex(),
                   "belt_mac_keyA17": r_mac(big, H[128:160]).hex()}, f, indent=1)
    from collections import Counter
    print("edge codes:", Counter(e["code"] for e in edge))
    print("golden written to", GOLD)


if __name__ == "__main__":
    if not refgen.have_ref():
        raise SystemExit("oracle/_ref/libbee2ref.so missing: run `make -C oracle ref` in the build container")
    main()
