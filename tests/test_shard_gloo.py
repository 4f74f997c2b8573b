"""CPU tests of the N>1 path: world_size-2 gloo.  Each rank processes its index shard (the
oracle stands in for the kernels -- no GPU here); the concatenation must equal the
single-stream result and the parameter block must arrive by broadcast."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    import torch
    import torch.distributed as dist

    import orclib
    from bee2_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = orclib.load()
    H = orc.beltH()
    # parameters exist on rank 0 only and travel once
    blob = b"".join(orc.ctr_start(H[128:160], H[192:208])) if rank == 0 else bytes(48)
    blob = shard.broadcast_params(dist, blob)
    kw, c0 = blob[:32], blob[32:]
    # CTR: ragged stream, rank-local counter offset
    nbytes = 16 * 1001 + 5
    stream = orc.fill(nbytes, 0xBE17)
    lo, hi, first = shard.ctr_shard(rank, world, nbytes)
    part = np.frombuffer(stream[lo:hi], dtype=np.uint8).copy()
    full_blocks = (hi - lo) // 16 * 16
    head = part[:full_blocks].copy()
    orc.ctr_blocks_np(head, kw, c0, first=first)
    out = head.tobytes()
    if hi - lo > full_blocks:          # ragged tail on the last rank: one more gamma block
        tail = np.zeros(16, dtype=np.uint8)
        tail[: hi - lo - full_blocks] = part[full_blocks:]
        orc.ctr_blocks_np(tail, kw, c0, first=first + full_blocks // 16)
        out += tail.tobytes()[: hi - lo - full_blocks]
    # bashF: independent states
    n = 1001
    states = orc.fill(192 * n, 0xBA5F)
    slo, shi = shard.shard_range(rank, world, n)
    bout = orc.bashF_batch(states[192 * slo: 192 * shi])
    # gather for the check (results only; nothing like this is on the data path)
    sizes = [None] * world
    dist.all_gather_object(sizes, (lo, hi, slo, shi))
    t = torch.tensor([len(out), len(bout)])
    dist.all_reduce(t)
    with open(os.path.join(outdir, f"r{rank}.bin"), "wb") as f:
        f.write(out + bout)
    with open(os.path.join(outdir, f"r{rank}.meta"), "w") as f:
        f.write(f"{len(out)} {len(bout)} {int(t[0])} {int(t[1])} {sizes}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ctr_and_bashF_equal_single_stream(tmp_path, orc):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    H = orc.beltH()
    nbytes = 16 * 1001 + 5
    stream = orc.fill(nbytes, 0xBE17)
    want_ctr = orc.ctr(stream, H[128:160], H[192:208])
    states = orc.fill(192 * 1001, 0xBA5F)
    want_bash = orc.bashF_batch(states)
    got_ctr, got_bash = b"", b""
    for r in range(world):
        raw = open(tmp_path / f"r{r}.bin", "rb").read()
        meta = open(tmp_path / f"r{r}.meta").read().split()
        a, b = int(meta[0]), int(meta[1])
        assert int(meta[2]) == nbytes and int(meta[3]) == 192 * 1001      # all-reduce saw every item
        got_ctr += raw[:a]
        got_bash += raw[a:a + b]
    assert got_ctr == want_ctr
    assert got_bash == want_bash


POLY = (1 << 128) | 0x87
Q_INV = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF82


def _gf_mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        if a >> 128:
            a ^= POLY
        b >>= 1
    return r


def _x_pow(j):
    p, b = 1, 2
    while j:
        if j & 1:
            p = _gf_mul(p, b)
        b = _gf_mul(b, b)
        j >>= 1
    return p


def _stream_worker(rank, world, port, outdir):
    """belt-bde / belt-che streams sharded by block index: rank r jumps the tweak / keystream state to its first
    block (what `first_block` makes the kernels do) and processes [lo, hi) with no data from the other rank"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist

    import orclib
    from bee2_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = orclib.load()
    H = orc.beltH()
    key = H[128:160]
    # rank 0 derives s = E_K(iv) and broadcasts it with the expanded key (the only exchange)
    blob = (bytes(orc.key_expand(key)) + orc.block_encr(H[192:208], key)) if rank == 0 else bytes(48)
    blob = shard.broadcast_params(dist, blob)
    s0 = int.from_bytes(blob[32:48], "little")
    nblocks = 1003
    data = orc.fill(16 * nblocks, 0xBDE)
    lo, hi = shard.shard_range(rank, world, nblocks)
    P = _x_pow(lo)
    s_bde = _gf_mul(s0, P)                                              # s * x^lo
    s_che = _gf_mul(s0, P) ^ _gf_mul(P ^ 1, Q_INV)                      # S_lo of s <- s*x ^ 1
    piece = data[16 * lo:16 * hi]
    bde, _ = orc.bde_blocks_from(piece, key, s_bde.to_bytes(16, "little"))
    che, _ = orc.che_blocks_from(piece, key, s_che.to_bytes(16, "little"))
    with open(os.path.join(outdir, f"s{rank}.bin"), "wb") as f:
        f.write(bde + che)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_bde_che_streams_equal_single_stream(tmp_path, orc):
    world = 2
    mp.spawn(_stream_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    H = orc.beltH()
    key, iv = H[128:160], H[192:208]
    data = orc.fill(16 * 1003, 0xBDE)
    from bee2_amd import shard
    got_bde, got_che = b"", b""
    for r in range(world):
        lo, hi = shard.shard_range(r, world, 1003)
        raw = open(tmp_path / f"s{r}.bin", "rb").read()
        n = 16 * (hi - lo)
        got_bde += raw[:n]
        got_che += raw[n:2 * n]
    assert got_bde == orc.bde(data, key, iv)[1]
    assert got_che == orc.dwp_wrap(data, b"", key, iv, "CHE")[1]


def test_shard_ranges_cover_and_balance():
    from bee2_amd import shard
    for n in (0, 1, 7, 8, 1000, 2 ** 20 + 3):
        for world in (1, 2, 3, 8):
            cuts = [shard.shard_range(r, world, n) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(2, 2, 10)
    lo, hi, first = shard.ctr_shard(1, 2, 16 * 10 + 3)
    assert (lo, hi, first) == (80, 163, 5)


def _sigvfy_worker(rank, world, port, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import json
    import torch
    import torch.distributed as dist

    import goldenlib
    import orclib
    from bee2_amd import shard
    from bee2_amd.engine import LEVEL_OID

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = orclib.load()
    items = goldenlib.Golden().sigvfy_pipeline["128"]
    lo, hi = shard.shard_range(rank, world, len(items))
    out = []
    for it in items[lo:hi]:                       # this rank's files: hash, validate the key, verify
        msg, pub, sig = (bytes.fromhex(it[k]) for k in ("msg", "pubkey", "sig"))
        dig = orc.belt_hash(msg)
        out.append((dig.hex(), orc.pubkey_val(128, pub), orc.verify_l(128, LEVEL_OID[128], dig, sig, pub)))
    bad = torch.tensor([sum(1 for _, k, v in out if k or v)])
    dist.all_reduce(bad)                          # the only cross-rank traffic such a job needs: a count
    with open(os.path.join(outdir, f"v{rank}.json"), "w") as f:
        json.dump({"lo": lo, "hi": hi, "out": out, "bad_total": int(bad[0])}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sigvfy_pipeline_equals_fixture(tmp_path, golden):
    """hash -> key validation -> verification over index shards: no exchange step, the
    concatenated verdicts are the reference's and the all-reduced failure count is the global one"""
    import json
    world = 2
    mp.spawn(_sigvfy_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    items = golden.sigvfy_pipeline["128"]
    got, pos = [], 0
    for r in range(world):
        d = json.load(open(tmp_path / f"v{r}.json"))
        assert d["lo"] == pos
        pos = d["hi"]
        got += d["out"]
        assert d["bad_total"] == sum(1 for it in items if it["pubkey_val"] or it["verify"])
    assert pos == len(items)
    assert [tuple(g) for g in got] == [(it["digest"], it["pubkey_val"], it["verify"]) for it in items]
