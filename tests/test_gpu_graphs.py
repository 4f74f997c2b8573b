"""-m gpu: the device-pointer entries are plain stream work -- kernels, memsets, a fork / join over the library's side stream -- so
once their scratch exists (first call) a caller can CAPTURE them into a hipGraph and replay it: bashF batch, beltCTR, ECB, the
fused bash512 + beltMAC kernel, a ragged belt-hash batch on two queues and a bign verification pipeline in ONE graph, replayed on
fresh inputs and compared with the eager results and the oracle."""
import numpy as np
import pytest
import torch

from gpulib import engine

pytestmark = pytest.mark.gpu


def test_device_entries_replay_from_one_hip_graph(orc, golden):
    eng = engine()
    H = golden.H
    kw, c0 = orc.ctr_start(H[128:160], H[192:208])
    key = H[128:160]
    n_states, n_blocks, n_msgs, ml = 4096, 1 << 16, 2048, 256
    states = torch.empty(192 * n_states, dtype=torch.uint8, device="cuda")
    ctr = torch.empty(16 * n_blocks, dtype=torch.uint8, device="cuda")
    ecb = torch.empty(16 * n_blocks, dtype=torch.uint8, device="cuda")
    msgs = torch.empty(n_msgs * ml, dtype=torch.uint8, device="cuda")
    dig = torch.empty(n_msgs * 64, dtype=torch.uint8, device="cuda")
    tag = torch.empty(n_msgs * 8, dtype=torch.uint8, device="cuda")
    # ragged batch: 1500 messages (>= 1024: long chains and short messages on two queues), a few long ones
    rnd = np.random.default_rng(5)
    lens = rnd.choice([0, 5, 32, 33, 200, 999], size=1500).astype(np.int64)
    lens[rnd.choice(1500, 6, replace=False)] = [4096, 5000, 9001, 4097, 16384, 7000]
    offs = np.zeros(1501, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    rdata = torch.empty(int(offs[-1]) + 16, dtype=torch.uint8, device="cuda")
    roff = torch.from_numpy(offs).cuda()
    rdig = torch.empty(1500 * 32, dtype=torch.uint8, device="cuda")
    # verification: the first 1024 genuine triples, a few damaged per round
    hs, ss, ps = golden.bign_base_arrays()
    nv = 1024
    vh = torch.empty(32 * nv, dtype=torch.uint8, device="cuda")
    vs = torch.empty(48 * nv, dtype=torch.uint8, device="cuda")
    vp = torch.empty(64 * nv, dtype=torch.uint8, device="cuda")
    codes = torch.empty(nv, dtype=torch.int32, device="cuda")
    bufs = (states, ctr, ecb, msgs, rdata)

    def fill(seed):
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        for b in bufs:
            b[: b.numel() // 8 * 8].view(torch.int64).random_(generator=g)
        S = np.frombuffer(ss, dtype=np.uint8)[: 48 * nv].reshape(nv, 48).copy()
        bad = np.random.default_rng(seed).choice(nv, 7, replace=False)
        S[bad, 3] ^= 0x10
        vh.copy_(torch.from_numpy(np.frombuffer(hs, dtype=np.uint8)[: 32 * nv].copy()))
        vs.copy_(torch.from_numpy(S.reshape(-1)))
        vp.copy_(torch.from_numpy(np.frombuffer(ps, dtype=np.uint8)[: 64 * nv].copy()))
        return [b.clone() for b in bufs], sorted(int(i) for i in bad)

    def work():
        eng.bashF_batch_dev(states)
        eng.beltCTR_blocks_dev(ctr, kw, c0, 0)
        eng.beltModes_blocks_dev(0, ecb, ecb, kw)
        eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, dig, tag, n_msgs)
        eng.hash_ragged_dev(0, rdata, roff, rdig, 1500)
        eng.bign128Verify_batch_dev(vh, vs, vp, codes)

    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        fill(1)
        work()                                   # first call: scratch, tables, dynamic-LDS grants
    cap.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        work()
    for seed in (11, 12, 13):
        with torch.cuda.stream(cap):
            plain, bad = fill(seed)
        cap.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        got = [t.cpu().numpy().tobytes() for t in (states, ctr, ecb, dig, tag, rdig)]
        got_codes = [int(c) & 0xFFFFFFFF for c in codes.cpu().numpy()]
        p_states, p_ctr, p_ecb, p_msgs, p_rdata = (t.cpu().numpy() for t in plain)
        assert got[0] == orc.bashF_batch(p_states.tobytes()), seed
        want = p_ctr.copy()
        orc.ctr_blocks_np(want, kw, c0, first=0, nthreads=8)
        assert got[1] == want.tobytes(), seed
        assert got[2] == orc.ecb(p_ecb.tobytes(), key)[1], seed
        mb = p_msgs.tobytes()
        for i in range(0, n_msgs, 97):
            m = mb[i * ml:(i + 1) * ml]
            assert got[3][64 * i: 64 * i + 64] == orc.bashHash(256, m)[1] and got[4][8 * i: 8 * i + 8] == orc.mac(m, key), (seed, i)
        rb = p_rdata.tobytes()
        for i in range(1500):
            assert got[5][32 * i: 32 * i + 32] == orc.belt_hash(rb[offs[i]:offs[i + 1]]), (seed, i)
        assert [i for i, c in enumerate(got_codes) if c] == bad and all(got_codes[i] in (505, 510) for i in bad), seed
