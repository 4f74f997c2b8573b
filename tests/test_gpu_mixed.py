"""-m gpu: per-message bashHash + beltMAC batches (H4) against the oracle."""
import numpy as np
import pytest
import torch

from gpulib import dev, engine, host

pytestmark = pytest.mark.gpu


def _check(eng, orc, key, msg_len, n, l, want_hash=True, want_mac=True, seed=1):
    msgs = orc.fill(n * msg_len, seed) if n * msg_len else b""
    dig, tag = eng.bashHash_beltMAC_batch(msgs, msg_len, l, key, want_hash, want_mac, n=n)
    for i in range(n):
        m = msgs[i * msg_len:(i + 1) * msg_len]
        if want_hash:
            assert dig[i * (l // 4):(i + 1) * (l // 4)] == orc.bashHash(l, m)[1], (msg_len, n, l, i)
        if want_mac:
            assert tag[8 * i: 8 * i + 8] == orc.mac(m, key), (msg_len, n, l, i)


@pytest.mark.parametrize("l", [128, 192, 256])
@pytest.mark.parametrize("msg_len", [0, 16, 48, 64, 80, 96, 128, 192, 256, 1024, 4096, 4096 + 48])
def test_fused_kernel_shapes(orc, golden, l, msg_len):
    eng = engine()
    _check(eng, orc, golden.H[128:160], msg_len, 67, l)


@pytest.mark.parametrize("msg_len", [1, 13, 15, 17, 100, 191, 193, 1000])
def test_generic_kernel_ragged_lengths(orc, golden, msg_len):
    eng = engine()
    _check(eng, orc, golden.H[128:160], msg_len, 33, 256)
    _check(eng, orc, golden.H[160:176], msg_len, 5, 64)          # 128-bit key, low level


def test_hash_only_and_mac_only(orc, golden):
    eng = engine()
    _check(eng, orc, golden.H[128:152], 4096, 130, 256, want_mac=False)   # 192-bit key unused
    _check(eng, orc, golden.H[128:152], 4096, 130, 256, want_hash=False)
    _check(eng, orc, golden.H[128:152], 100, 9, 256, want_hash=False)


def test_batch_sizes(orc, golden):
    eng = engine()
    for n in (1, 63, 64, 65, 1023, 1025, 3000):
        _check(eng, orc, golden.H[128:160], 256, n, 256, seed=n)


def test_mixed_per_gpu_share_of_config4(orc, golden):
    """BASELINE.json configs[4] is 2^24 x 4 KiB over 8 GPUs = 2^21 messages per GPU.  Run one GPU's share from
    HBM-resident data and compare EVERY digest and tag with the oracle (all host cores, 2^16 messages at a
    time), then check idempotence (same input -> same outputs on a second pass)."""
    import os
    eng = engine()
    n, msg_len = 1 << 21, 4096
    free, _ = torch.cuda.mem_get_info()
    if free < n * msg_len + (1 << 30):
        pytest.skip("not enough HBM free")
    key = golden.H[128:160]
    msgs = torch.empty(n * msg_len, dtype=torch.uint8, device="cuda")
    msgs.view(torch.int64).random_()
    dig = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    tag = torch.zeros(n * 8, dtype=torch.uint8, device="cuda")
    eng.bashHash_beltMAC_batch_dev(msgs, msg_len, 256, key, dig, tag)
    torch.cuda.synchronize()
    d1, t1 = dig.clone(), tag.clone()
    threads = os.cpu_count() or 8
    step = 1 << 16
    hd, ht = host(dig), host(tag)
    for lo in range(0, n, step):
        m = host(msgs[lo * msg_len:(lo + step) * msg_len])
        wd, wt = orc.mixed_batch(m, msg_len, key, nthreads=threads)
        assert hd[64 * lo: 64 * (lo + step)] == wd, f"digests of messages {lo}..{lo + step}"
        assert ht[8 * lo: 8 * (lo + step)] == wt, f"tags of messages {lo}..{lo + step}"
    dig.zero_(); tag.zero_()
    eng.bashHash_beltMAC_batch_dev(msgs, msg_len, 256, key, dig, tag)
    torch.cuda.synchronize()
    assert torch.equal(dig, d1) and torch.equal(tag, t1)


def test_config4_whole_job_on_one_gpu(orc, golden):
    """BASELINE.json configs[4] as ONE job: 2^24 x 4 KiB = 64 GiB resident on one card, one
    bee2hip_bashHash_beltMAC_batch_dev call (the reference's per-message loop: bash_hash.c:38-137, belt_mac.c:47-203;
    VERDICT r04 item 4).  The byte offsets of the messages pass 2^32, 2^35 and end at 2^36: windows of 2^12 messages at the
    start, either side of bytes 2^32 and 2^35 and at the end are compared with the oracle, the outputs in between are checked to have
    been written, and a guard behind each output array must survive."""
    import os
    eng = engine()
    n, msg_len, W = 1 << 24, 4096, 1 << 12
    free, _ = torch.cuda.mem_get_info()
    if free < 80 << 30:
        pytest.skip("needs 80 GiB of free HBM")
    key = golden.H[128:160]
    msgs = torch.empty(n * msg_len, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(0x4D1C)
    for lo in range(0, n * msg_len, 1 << 33):                       # (8 GiB at a time: the generator's counter)
        msgs[lo: lo + (1 << 33)].view(torch.int64).random_(generator=g)
    GUARD = 4096
    dig = torch.full((n * 64 + GUARD,), 0xA5, dtype=torch.uint8, device="cuda")
    tag = torch.full((n * 8 + GUARD,), 0x5A, dtype=torch.uint8, device="cuda")
    eng.bashHash_beltMAC_batch_dev(msgs, msg_len, 256, key, dig[: n * 64], tag[: n * 8], n=n)
    torch.cuda.synchronize()
    assert bool((dig[n * 64:] == 0xA5).all()) and bool((tag[n * 8:] == 0x5A).all()), "wrote behind the outputs"
    threads = len(os.sched_getaffinity(0))
    windows = [0, (1 << 20) - W // 2, (1 << 23) - W // 2, n - W]     # message 2^20 starts at byte 2^32, message 2^23 at byte 2^35
    for lo in windows:
        m = host(msgs[lo * msg_len:(lo + W) * msg_len])
        wd, wt = orc.mixed_batch(m, msg_len, key, nthreads=threads)
        assert host(dig[64 * lo: 64 * (lo + W)]) == wd, f"digests of messages {lo}..{lo + W}"
        assert host(tag[8 * lo: 8 * (lo + W)]) == wt, f"tags of messages {lo}..{lo + W}"
    # every message got outputs: no 64-byte digest / 8-byte tag still holds the fill pattern (probability 2^-64 per tag otherwise)
    assert not bool((dig[: n * 64].view(n, 64) == 0xA5).all(dim=1).any())
    assert not bool((tag[: n * 8].view(n, 8) == 0x5A).all(dim=1).any())
    # and the windows agree with a second, separate call over just that window (the kernel's indexing does not depend on the batch)
    for lo in windows[1:]:
        d2 = torch.empty(W * 64, dtype=torch.uint8, device="cuda")
        t2 = torch.empty(W * 8, dtype=torch.uint8, device="cuda")
        eng.bashHash_beltMAC_batch_dev(msgs[lo * msg_len:(lo + W) * msg_len], msg_len, 256, key, d2, t2, n=W)
        torch.cuda.synchronize()
        assert torch.equal(d2, dig[64 * lo: 64 * (lo + W)]) and torch.equal(t2, tag[8 * lo: 8 * (lo + W)])


def test_ragged_hash_batches(orc, golden):
    """messages of different lengths in one launch, vs the oracle and the golden set"""
    import random
    eng = engine()
    rnd = random.Random(21)
    msgs = [bytes.fromhex(c["msg"]) for c in golden.belt_bash]
    msgs += [orc.fill(n, n) for n in (0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 5000)]
    # around the switch to 8 lanes per message (4096 bytes) and around its rate blocks (64 / 96 / 128 bytes)
    msgs += [orc.fill(n, n) for n in (4095, 4096, 4097, 4096 + 63, 4096 + 64, 4096 + 95, 4096 + 96, 4096 + 127,
                                      4096 + 128, 4096 + 129, 8192, 65536 + 5, 300_000)]
    msgs += [rnd.randbytes(rnd.randrange(0, 3000)) for _ in range(200)]
    for alg in (0, 128, 192, 256):
        code, digs = eng.hash_ragged(alg, msgs)
        assert code == 0
        for m, d in zip(msgs, digs):
            want = orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1]
            assert d == want, (alg, len(m))
    for c, d in zip(golden.belt_bash, eng.hash_ragged(256, [bytes.fromhex(c["msg"]) for c in golden.belt_bash])[1]):
        assert d.hex() == c["bash512"]
    assert eng.hash_ragged(100, [b"x"])[0] == 502


def test_belt_hash_dropin_A23_and_streaming(orc, golden):
    """beltHash{Start,StepH,StepG,StepG2,StepV,StepV2} / beltHash (belt_test.c:593-627, STB A.23): one-shot on
    the STB vectors and on ragged lengths, then arbitrary splits with a digest taken after every piece"""
    import random
    eng = engine()
    H = golden.H
    for k in golden.kat["belt_hash"]:
        assert eng.beltHash(H[: k["len"]]) == (0, bytes.fromhex(k["out"])), k["name"]
    rnd = random.Random(41)
    for n in (0, 1, 31, 32, 33, 63, 64, 65, 1000, 4096, 100_003):
        msg = orc.fill(n, n + 5)
        assert eng.beltHash(msg) == (0, orc.belt_hash(msg)), n
        for _ in range(2):
            splits, left = [], n
            while left:
                s = min(left, rnd.choice((1, 5, 31, 32, 33, 64, 100, 5000)))
                splits.append(s)
                left -= s
            splits = splits or [0]
            outs = eng.beltHash_steps(msg, splits, hash_len=rnd.choice((32, 16, 7)))
            off = 0
            for s_, d in zip(splits, outs):                       # the digest of every prefix, truncated
                off += s_
                assert d == orc.belt_hash(msg[:off])[: len(d)], (n, off)
    for c in golden.belt_bash[:8]:
        assert eng.beltHash(bytes.fromhex(c["msg"]))[1].hex() == c["belt_hash"]


def test_ragged_hash_device_api_with_launch_order(orc):
    """bee2hip_hash_ragged[_ordered]_dev: digests land at the caller's index whatever the launch order
    (identity, longest-first, a random permutation); unaligned message starts included"""
    import random
    import numpy as np
    import torch
    eng = engine()
    rnd = random.Random(77)
    msgs = [rnd.randbytes(rnd.choice((0, 1, 5, 31, 32, 33, 127, 128, 129, 1000, 4097))) for _ in range(300)]
    offs = np.zeros(len(msgs) + 1, dtype=np.int64)
    np.cumsum([len(m) for m in msgs], out=offs[1:])
    blob = b"".join(msgs)
    data = torch.from_numpy(np.frombuffer(blob + b"\0" * 16, dtype=np.uint8).copy()).cuda()
    doff = torch.from_numpy(offs).cuda()
    n = len(msgs)
    lens = np.diff(offs)
    orders = {"identity": None,
              "longest_first": np.argsort(-lens, kind="stable").astype(np.int32),
              "random": np.array(rnd.sample(range(n), n), dtype=np.int32)}
    for alg, dl in ((0, 32), (128, 32), (192, 48), (256, 64)):
        want = [orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1] for m in msgs]
        for name, o in orders.items():
            dig = torch.zeros(n * dl, dtype=torch.uint8, device="cuda")
            eng.hash_ragged_dev(alg, data, doff, dig, n, order=None if o is None else torch.from_numpy(o).cuda())
            eng.sync()
            got = dig.cpu().numpy().tobytes()
            for i in range(n):
                assert got[i * dl:(i + 1) * dl] == want[i], (alg, name, i, len(msgs[i]))


def test_bsum_front_end_example(orc, golden, tmp_path):
    """build examples/bsum_hip.c against the C ABI and compare its output with bsum's format"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("cc") is None:
        pytest.skip("no C compiler on this box")
    exe = tmp_path / "bsum_hip"
    lib = os.path.join(root, "bee2_amd", "lib")
    subprocess.check_call(["cc", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "bsum_hip.c"),
                           "-L" + lib, "-lbee2hip", "-Wl,-rpath," + lib, "-o", str(exe)])
    names = []
    for i, n in enumerate((0, 1, 100, 4096, 70000)):
        p = tmp_path / f"f{i}.bin"
        p.write_bytes(orc.fill(n, 77 + i))
        names.append(str(p))
    for flag, alg in (("-belt-hash", 0), ("-bash256", 128), ("-bash512", 256)):
        out = subprocess.check_output([str(exe), flag] + names, text=True).splitlines()
        for line, name in zip(out, names):
            data = open(name, "rb").read()
            want = orc.belt_hash(data) if alg == 0 else orc.bashHash(alg, data)[1]
            assert line == f"{want.hex()}  {name}"             # lower case, two spaces: bsumPrint (bsum.c:207-224)
        # check mode (bsumCheck, bsum.c:226-306): all good -> exit 0 and "name: OK" per line
        sums = tmp_path / f"sums{alg}.txt"
        sums.write_text("\n".join(out) + "\n")
        r = subprocess.run([str(exe), flag, "-c", str(sums)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.splitlines() == [f"{n}: OK" for n in names] and r.stderr == ""
        # one wrong digest, one missing file, one malformed line, CRLF line ends, upper-case hex
        lines = list(out)
        lines[1] = ("0" if lines[1][0] != "0" else "1") + lines[1][1:]
        lines[2] = lines[2].split("  ")[0].upper() + "  " + names[2] + "\r"
        lines.append(lines[0].split("  ")[0] + "  " + str(tmp_path / "missing.bin"))
        lines.append("not a checksum line")
        sums.write_text("\n".join(lines) + "\n")
        r = subprocess.run([str(exe), flag, "-c", str(sums)], capture_output=True, text=True)
        assert r.returncode != 0
        # the missing file is reported while the list is read, the verdicts after the single launch
        assert sorted(r.stdout.splitlines()) == sorted([f"{names[0]}: OK", f"{names[1]}: FAILED [checksum]", f"{names[2]}: OK", f"{names[3]}: OK",
                                                        f"{names[4]}: OK", f"{tmp_path / 'missing.bin'}: FAILED [open]"])
        assert r.stderr.splitlines() == ["WARNING: 1 input line (out of 7) is improperly formatted",
                                         "WARNING: 1 listed file could not be opened or read",
                                         "WARNING: 1 computed checksum did not match"]


def test_sigvfy_front_end_example(orc, golden, tmp_path):
    """examples/sigvfy_hip.c: the batch shape of `bee2cmd sig vfy` (hash the file, bign128PubkeyVal, bign128Verify) from
    plain C on device-resident buffers; files signed here with the library's own bign128Sign2, verdicts per line"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("cc") is None:
        pytest.skip("no C compiler on this box")
    exe = tmp_path / "sigvfy_hip"
    lib = os.path.join(root, "bee2_amd", "lib")
    subprocess.check_call(["cc", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "sigvfy_hip.c"),
                           "-L" + lib, "-lbee2hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    eng = engine()
    lines, want = [], []
    for i, n in enumerate((0, 1, 100, 4096, 70000, 33)):
        p = tmp_path / f"s{i}.bin"
        data = orc.fill(n, 500 + i)
        p.write_bytes(data)
        d = orc.fill(32, 900 + i)
        code, pub = eng.bignLPubkeyCalc(128, d)
        assert code == 0
        code, sig = eng.bignLSign2(128, orc.belt_hash(data), d)
        assert code == 0 and orc.verify(orc.belt_hash(data), sig, pub) == 0
        verdict = "OK"
        if i == 2:
            sig = bytes([sig[0] ^ 1]) + sig[1:]
            verdict = "FAILED [signature]"
        if i == 4:
            pub = pub[:32] + bytes([pub[32] ^ 1]) + pub[33:]
            verdict = "FAILED [pubkey]"
        lines.append(f"{p} {sig.hex()} {pub.hex()}")
        want.append(f"{p}: {verdict}")
    lines.append(f"{tmp_path / 'nofile.bin'} {'00' * 48} {'00' * 64}")
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(lines) + "\n")
    r = subprocess.run([str(exe), str(lst)], capture_output=True, text=True)
    assert r.returncode == 1
    assert sorted(r.stdout.splitlines()) == sorted(want + [f"{tmp_path / 'nofile.bin'}: FAILED [open]"])
    lst.write_text("\n".join(l for l, w in zip(lines, want) if w.endswith("OK")) + "\n")
    r = subprocess.run([str(exe), str(lst)], capture_output=True, text=True)
    assert r.returncode == 0 and all(x.endswith(": OK") for x in r.stdout.splitlines()) and len(r.stdout.splitlines()) == 4
    # a list under ONE key: the example takes the one-signer entry (bee2hip_bignVerifyL_onekey_batch_dev)
    d = orc.fill(32, 0x1516)
    code, pub = eng.bignLPubkeyCalc(128, d)
    assert code == 0
    lines, want = [], []
    for i, n in enumerate((5, 0, 3000, 77, 64)):
        p = tmp_path / f"k{i}.bin"
        data = orc.fill(n, 700 + i)
        p.write_bytes(data)
        code, sig = eng.bignLSign2(128, orc.belt_hash(data), d)
        assert code == 0
        if i == 3:
            sig = sig[:20] + bytes([sig[20] ^ 8]) + sig[21:]
        lines.append(f"{p} {sig.hex()} {pub.hex()}")
        want.append(f"{p}: {'FAILED [signature]' if i == 3 else 'OK'}")
    lst.write_text("\n".join(lines) + "\n")
    r = subprocess.run([str(exe), str(lst)], capture_output=True, text=True)
    assert r.returncode == 1 and sorted(r.stdout.splitlines()) == sorted(want), (r.stdout, r.stderr)


def test_c_selftest_example_runs_the_stb_vectors(golden, tmp_path):
    """examples/selftest.c: the drop-in functions from plain C (what a bee2 maintainer would try first), on the
    STB vectors of the fixtures: bash A.2 / A.3, belt A.9-A.12, A.15-A.17, A.19-A.20, A.23-A.25, bign G.2 / G.3"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("cc") is None:
        pytest.skip("no C compiler on this box")
    exe = tmp_path / "selftest"
    lib = os.path.join(root, "bee2_amd", "lib")
    subprocess.check_call(["cc", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "selftest.c"),
                           "-L" + lib, "-lbee2hip", "-Wl,-rpath," + lib, "-o", str(exe)])
    k, H = golden.kat, golden.H
    hx = lambda b: b.hex() if len(b) else "-"
    lines = [f"bashF A.2 {k['bashF_A2']['in']} {k['bashF_A2']['out']}"]
    lines += [f"bashhash {v['name']} {v['l']} {hx(H[:v['len']])} {v['out']}" for v in k["bash_hash"]]
    lines += [f"belthash {v['name']} {hx(H[:v['len']])} {v['out']}" for v in k["belt_hash"]]
    lines += [f"ctr {v['name']} {v['in']} {v['key']} {v['iv']} {v['out']}" for v in k["belt_ctr"]]
    lines += [f"mac {v['name']} {v['in']} {v['key']} {v['out']}" for v in k["belt_mac"]]
    lines += [f"mode {v['name']} {v['fn']} {v['in']} {v['key']} {v['iv'] or '-'} {v['out']}" for v in k["belt_modes"]]
    for mode, g in (("DWP", golden.belt_dwp), ("CHE", golden.belt_che)):
        v = g["kat"][0]
        assert v["op"] == "wrap"
        lines.append(f"wrap {v['name']} {mode} {v['crit']} {v['open']} {v['key']} {v['iv']} {v['out']} {v['mac']}")
        for i, c in enumerate(g["short"][:6]):
            lines.append(f"wrap short{i} {mode} {c['crit'] or '-'} {c['open'] or '-'} {c['key']} {c['iv']} {c['out'] or '-'} {c['mac']}")
    lines += [f"verify {v['name']} {v['hash']} {v['sig']} {v['pubkey']} {v['code']}" for v in k["bign_verify"]]
    vec = tmp_path / "vectors.txt"
    vec.write_text("\n".join(lines) + "\n")
    r = subprocess.run([str(exe), str(vec)], capture_output=True, text=True, timeout=300)
    out = r.stdout.strip().splitlines()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert not [l for l in out if l.startswith(("FAIL", "SKIP"))]
    assert out[-1].startswith(f"{len(lines)} vectors, 0 failed"), out[-1]


@pytest.mark.parametrize("l", [128, 192, 256])
def test_sigvfy_pipeline_on_one_stream(golden, l):
    """`bee2cmd sig vfy` as a batch (cmd/core/cmd_sig.c:461-490: hash the file, bignPubkeyVal, bignVerify): the
    ragged hash writes its digests where the verification reads its hashes, everything queued on one stream with
    no host round trip in between; verdicts are the reference's (tests/golden/sigvfy_pipeline.json)."""
    import numpy as np
    from bee2_amd import engine as E
    eng = engine()
    items = golden.sigvfy_pipeline[str(l)] * 30                      # ~2 000 messages, several wavefronts
    msgs = [bytes.fromhex(it["msg"]) for it in items]
    offs = np.zeros(len(msgs) + 1, dtype=np.int64)
    np.cumsum([len(m) for m in msgs], out=offs[1:])
    data = dev(b"".join(msgs) + bytes(16))
    doff = torch.from_numpy(offs).cuda()
    sigs = dev(b"".join(bytes.fromhex(it["sig"]) for it in items))
    pubs = dev(b"".join(bytes.fromhex(it["pubkey"]) for it in items))
    n = len(items)
    digests = torch.zeros(n * (l // 4), dtype=torch.uint8, device="cuda")
    kcodes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    vcodes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.hash_ragged_dev(0 if l == 128 else l, data, doff, digests, n)
        eng.bignPubkeyValL_batch_dev(l, pubs, kcodes)
        eng.bignVerifyL_batch_dev(l, E.LEVEL_OID[l], digests, sigs, pubs, vcodes)
    st.synchronize()
    assert host(digests) == b"".join(bytes.fromhex(it["digest"]) for it in items)
    assert [int(c) & 0xFFFFFFFF for c in kcodes.cpu().numpy()] == [it["pubkey_val"] for it in items]
    assert [int(c) & 0xFFFFFFFF for c in vcodes.cpu().numpy()] == [it["verify"] for it in items]


@pytest.mark.parametrize("l", [128, 192, 256])
def test_sigvfy_pipeline_of_signers_in_one_keyed_call(golden, l):
    """the same pipeline with the signers as a SET: the fixture's distinct public keys once, an index per file, the keyed entry in the
    place of the general one -- the ragged hash, the key validation (of the few keys) and the verification on one stream; the
    reference's verdicts (tests/golden/sigvfy_pipeline.json)"""
    import numpy as np
    from bee2_amd import engine as E
    eng = engine()
    items = golden.sigvfy_pipeline[str(l)] * 30
    msgs = [bytes.fromhex(it["msg"]) for it in items]
    offs = np.zeros(len(msgs) + 1, dtype=np.int64)
    np.cumsum([len(m) for m in msgs], out=offs[1:])
    data = dev(b"".join(msgs) + bytes(16))
    doff = torch.from_numpy(offs).cuda()
    sigs = dev(b"".join(bytes.fromhex(it["sig"]) for it in items))
    keys = sorted({bytes.fromhex(it["pubkey"]) for it in items})
    kidx = torch.tensor([keys.index(bytes.fromhex(it["pubkey"])) for it in items], dtype=torch.int32).cuda()
    n = len(items)
    digests = torch.zeros(n * (l // 4), dtype=torch.uint8, device="cuda")
    kcodes = torch.full((len(keys),), -1, dtype=torch.int32, device="cuda")
    vcodes = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    dkeys = dev(b"".join(keys))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.hash_ragged_dev(0 if l == 128 else l, data, doff, digests, n)
        eng.bignPubkeyValL_batch_dev(l, dkeys, kcodes)                               # every signer once
        eng.bignVerifyL_keyed_batch_dev(l, E.LEVEL_OID[l], digests, sigs, b"".join(keys), kidx, vcodes)
    st.synchronize()
    assert host(digests) == b"".join(bytes.fromhex(it["digest"]) for it in items)
    kc = [int(c) & 0xFFFFFFFF for c in kcodes.cpu().numpy()]
    assert [kc[keys.index(bytes.fromhex(it["pubkey"]))] for it in items] == [it["pubkey_val"] for it in items]
    assert [int(c) & 0xFFFFFFFF for c in vcodes.cpu().numpy()] == [it["verify"] for it in items]


@pytest.mark.parametrize("n", [32767, 32768 + 3, 256 * 1024 + 5])
def test_ragged_belt_hash_many_short_messages_every_table_variant(orc, n):
    """The short-message belt-hash kernel switches table and workgroup shape with the batch size (4 KiB table /
    64 threads below 2^15 messages, 64 KiB table / 256 threads, 64 KiB / 1024 threads once every CU gets one):
    batches on each side, lengths 0..199 so every alignment and every partial-block length occurs, every
    digest against the oracle (sampled for the largest batch); bash256 on the same batch as well."""
    import random
    eng = engine()
    rnd = random.Random(n)
    lens = np.array([rnd.randrange(0, 200) for _ in range(n)], dtype=np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    blob = rnd.randbytes(int(offs[-1]))
    data = dev(blob + bytes(16))
    doff = torch.from_numpy(offs).cuda()
    step = 1 if n < 100000 else 7
    for alg, dl in ((0, 32), (128, 32)):
        dig = torch.zeros(n * dl, dtype=torch.uint8, device="cuda")
        eng.hash_ragged_dev(alg, data, doff, dig, n)
        torch.cuda.synchronize()
        got = host(dig)
        for i in range(0, n, step):
            m = blob[offs[i]:offs[i + 1]]
            want = orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1]
            assert got[dl * i: dl * i + dl] == want, (alg, n, i, len(m))


@pytest.mark.gpu
def test_host_pointer_ragged_batch_hands_its_few_giant_messages_to_host_threads(orc):
    """bee2hip_hash_ragged (HOST pointers): a message is one dependent chain, which a GPU lane walks 15x slower than a host
    core -- a batch of small messages with a few very long ones would wait tens of milliseconds for those chains.  The entry
    hashes the K longest on host threads while the GPU takes the rest (capi.hip); the digests are the oracle's either way,
    BEE2HIP_FORCE=gpu keeps everything on the device, and the device-pointer entry never offloads."""
    import ctypes
    import random
    import time

    import numpy as np

    from gpulib import engine
    eng = engine()
    L = eng.lib
    rnd = random.Random(77)
    lens = [rnd.randrange(0, 3000) for _ in range(3000)] + [1 << 20, (1 << 19) + 13, 300_001, 70_000, 65_536, 65_535]
    rnd.shuffle(lens)
    data = orc.fill(sum(lens), 0x4D1C)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    for alg, dlen in ((0, 32), (128, 32), (256, 64)):
        want = []
        for i, n in enumerate(lens):
            m = data[int(offs[i]): int(offs[i + 1])]
            want.append(orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1])
        times = {}
        for mode in (0, 1):
            L.bee2hip_path_policy(mode)
            out = ctypes.create_string_buffer(dlen * len(lens))
            s0 = L.bee2hip_path_count(0)
            t0 = time.perf_counter()
            assert L.bee2hip_hash_ragged(ctypes.c_size_t(alg), data, offs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(lens)), out) == 0
            times[mode] = time.perf_counter() - t0
            offloaded = L.bee2hip_path_count(0) - s0
            assert (offloaded == 1) == (mode == 0), (alg, mode, offloaded)
            for i in range(len(lens)):
                assert out.raw[dlen * i: dlen * (i + 1)] == want[i], (alg, mode, i, lens[i])
        L.bee2hip_path_policy(0)
        if alg == 0:
            assert times[0] < 0.5 * times[1], times          # the 1 MiB belt-hash chain alone is ~0.13 s on a lane pair


def test_ragged_batch_on_two_queues_keeps_the_callers_stream_order(orc):
    """bee2hip_hash_ragged_dev puts the long chains on the caller's stream and the short messages on the calling thread's side
    stream (batches of 1024 messages and more): the fork must wait for what the caller queued before the call (the upload of the
    messages), the join must hold back what the caller queues after it (here: the buffer is overwritten and the digests are
    copied out, all on one non-default stream, no synchronisation in between).  Digests against the oracle, three rounds, belt-hash
    and bash256; a batch under 1024 messages (one queue) beside it."""
    import random
    eng = engine()
    rnd = random.Random(4242)
    for n in (1500, 700):
        lens = [rnd.choice((0, 7, 32, 100, 999, 2000, 4095)) for _ in range(n)]
        for k in rnd.sample(range(n), 12):
            lens[k] = rnd.choice((4096, 5000, 20000, 70001))
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        total = int(offs[-1])
        doff = torch.from_numpy(offs).cuda()
        side = torch.cuda.Stream()
        for rnd_no, (alg, dl) in enumerate(((0, 32), (128, 32), (0, 32))):
            blob = rnd.randbytes(total)
            staged = torch.from_numpy(np.frombuffer(blob + bytes(16), dtype=np.uint8).copy()).pin_memory()
            data = torch.empty(total + 16, dtype=torch.uint8, device="cuda")
            dig = torch.zeros(n * dl, dtype=torch.uint8, device="cuda")
            out = torch.empty(n * dl, dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                data.copy_(staged, non_blocking=True)               # queued BEFORE the call
                eng.hash_ragged_dev(alg, data, doff, dig, n)
                data.fill_(0xA5)                                    # queued AFTER it: must not reach either kernel
                out.copy_(dig, non_blocking=True)
            side.synchronize()
            got = out.numpy().tobytes()
            for i in range(n):
                m = blob[offs[i]:offs[i + 1]]
                want = orc.belt_hash(m) if alg == 0 else orc.bashHash(alg, m)[1]
                assert got[dl * i: dl * i + dl] == want, (n, rnd_no, alg, i, len(m))
