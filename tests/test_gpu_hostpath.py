"""-m gpu: the drop-in layer's path policy (bee2_amd/csrc/staging.hpp "host path for small single calls").
Default mode: small single calls on the host, large ones and every batch entry point on the GPU; a device failure in the
middle of a void bee2 function is retried once and then finished on the host (never abort() in auto mode); with
BEE2HIP_FORCE=gpu the failure is fatal, as in rounds 1-2."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from bee2_amd import engine as E
from gpulib import engine, exp_engine

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats(lib):
    return [lib.bee2hip_path_count(i) for i in range(3)]


def test_auto_mode_crossovers(orc):
    eng = engine()
    L = eng.lib
    L.bee2hip_path_policy(0)
    H = orc.beltH()
    # one permutation, one block, a 16-byte CTR step: host
    s0 = _stats(L)
    st = ctypes.create_string_buffer(orc.fill(192, 1), 192)
    L.bashF(st, None)
    assert st.raw == orc.bashF(orc.fill(192, 1))
    cs = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(cs, H[128:160], ctypes.c_size_t(32), H[192:208])
    b = ctypes.create_string_buffer(bytes(16), 16)
    L.beltCTRStepE(b, ctypes.c_size_t(16), cs)
    assert b.raw == orc.ctr(bytes(16), H[128:160], H[192:208])
    s1 = _stats(L)
    assert s1[0] > s0[0] and s1[1] == s0[1]
    # a 64 KiB CTR call: GPU
    msg = orc.fill(1 << 16, 2)
    out = ctypes.create_string_buffer(1 << 16)
    assert L.beltCTR(out, msg, ctypes.c_size_t(1 << 16), H[128:160], ctypes.c_size_t(32), H[192:208]) == 0
    assert out.raw == orc.ctr(msg, H[128:160], H[192:208])
    s2 = _stats(L)
    assert s2[1] == s1[1] + 1                      # the bulk encryption (the start's E_K(iv) is a single block: host)
    # a serial chain of one message: host at every size
    big = orc.fill(1 << 20, 3)
    dig = ctypes.create_string_buffer(32)
    assert L.bashHash(dig, ctypes.c_size_t(128), big, ctypes.c_size_t(len(big))) == 0
    assert dig.raw == orc.bashHash(128, big)[1]
    assert L.beltHash(dig, big, ctypes.c_size_t(len(big))) == 0
    assert dig.raw == orc.belt_hash(big)
    s3 = _stats(L)
    assert s3[1] == s2[1] and s3[2] == 0
    # the batch entry point never takes the host path, whatever the size
    one = np.frombuffer(orc.fill(192, 4), dtype=np.uint8).copy()
    assert L.bee2hip_bashF_batch(ctypes.c_void_p(one.ctypes.data), ctypes.c_size_t(1)) == 0
    assert one.tobytes() == orc.bashF(orc.fill(192, 4))
    cs2 = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(cs2, H[128:160], ctypes.c_size_t(32), H[192:208])
    s4 = _stats(L)
    b2 = ctypes.create_string_buffer(bytes(48), 48)
    assert L.bee2hip_beltCTR_bulk(b2, ctypes.c_size_t(48), cs2) == 0
    assert b2.raw == orc.ctr(bytes(48), H[128:160], H[192:208])
    assert _stats(L)[0] == s4[0]


def test_one_signature_is_verified_on_the_host_batches_never(orc, golden):
    """bign128Verify / bign192Verify / bign256Verify / bignVerify on a standard curve: the calling core in auto mode
    (bee2_amd/csrc/host_bign.hpp), the GPU under BEE2HIP_FORCE=gpu, the same verdicts either way; the batch entry point
    stays on the GPU even for n = 1"""
    eng = engine()
    L = eng.lib
    cases = [tuple(bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey")) + (k["code"],) for k in golden.kat["bign_verify"]]
    cases += [tuple(bytes.fromhex(k[x]) for x in ("hash", "sig", "pubkey")) + (k["code"],) for k in golden.bign_edge[::7]]
    for mode in (0, 1, 2):
        L.bee2hip_path_policy(mode)
        s0 = _stats(L)
        for h, s, p, code in cases:
            assert eng.bign128Verify(h, s, p) == code
        s1 = _stats(L)
        assert (s1[0] - s0[0] == len(cases)) == (mode != 1), mode
    L.bee2hip_path_policy(0)
    for l in (192, 256):
        fn = lambda h, s, p: eng.bignLVerify(l, h, s, p)   # noqa: E731
        for e in golden.bign_big[str(l)]["edge"][::5] + golden.bign_big[str(l)]["base"][:8]:
            h, s, p = (bytes.fromhex(e[x]) for x in ("hash", "sig", "pubkey"))
            want = e.get("code", 0)
            s0 = _stats(L)
            assert fn(h, s, p) == want
            assert _stats(L)[0] == s0[0] + 1
            L.bee2hip_path_policy(1)
            assert fn(h, s, p) == want
            L.bee2hip_path_policy(0)
    # one public key: host in auto mode, GPU when forced, the reference's codes either way
    for l in (128, 192, 256):
        for c in golden.bign_pubkey_val[str(l)][::9]:
            pub = bytes.fromhex(c["pubkey"])
            s0 = _stats(L)
            assert eng.bignLPubkeyVal(l, pub) == c["code"], (l, c["name"])
            assert _stats(L)[0] == s0[0] + 1
            L.bee2hip_path_policy(1)
            assert eng.bignLPubkeyVal(l, pub) == c["code"], (l, c["name"])
            assert _stats(L)[0] == s0[0] + 1
            L.bee2hip_path_policy(0)
    # n = 1 through the batch entry point: GPU
    h, s, p, code = cases[0]
    s0 = _stats(L)
    rc, codes = eng.bignVerify_batch(h, s, p)
    assert rc == 0 and codes == [code] and _stats(L)[0] == s0[0]
    # a bad OID is still the batch entry's ERR_BAD_OID, a non-standard parameter set still goes to the general-curve kernels
    params = eng.bignParamsStd(E.CURVE_NAME[128])
    assert eng.bignVerify(params, b"\x06\x02\x2a", h, s, p) == 301
    assert _stats(L)[0] == s0[0]


def test_one_signature_is_made_on_the_host_in_auto_mode_batches_never(orc, golden):
    """bignSign2 / bignSign / bignPubkeyCalc / bignKeypairGen for ONE item on a standard curve: the calling core in auto
    mode (constant-time host arithmetic, bee2_amd/csrc/host_bign_ct.hpp), the GPU kernels under BEE2HIP_FORCE=gpu -- the same
    octets either way (the signatures are deterministic), and the batch entry point stays on the GPU even for n = 1"""
    eng = engine()
    L = eng.lib
    for l in (128, 192, 256):
        no = l // 4
        P = eng.bignParamsStd(E.CURVE_NAME[l])
        oid = E.LEVEL_OID[l]
        d = bytes(orc.fill(no - 1, 0x5160 + l)) + b"\x3f"
        h = orc.fill(no, 0x5161 + l)
        outs = {}
        for mode in (0, 1, 2):
            L.bee2hip_path_policy(mode)
            s0 = _stats(L)
            pub = eng.bignPubkeyCalc(P, d)
            sig = eng.bignSign2(P, oid, h, d, None)
            sig_t = eng.bignSign2(P, oid, h, d, b"additional input " * 9)          # 153 octets: beyond the nonce kernel's 64
            outs[mode] = (pub, sig, sig_t)
            s1 = _stats(L)
            assert (s1[0] - s0[0] >= 3) == (mode != 1), (l, mode, s0, s1)
        L.bee2hip_path_policy(0)
        assert outs[0] == outs[1] == outs[2]
        assert outs[0][0] == orc.pubkey_calc(l, d) and outs[0][1] == orc.sign2(l, oid, h, d, None)
        assert eng.bignLVerify(l, h, outs[0][1][1], outs[0][0][1]) == 0
        # n = 1 through the batch entry point: GPU
        s0 = _stats(L)
        code, sigs, codes = eng.bignSign2_batch(P, oid, h, d, None)
        assert code == 0 and codes == [0] and sigs == outs[0][1][1] and _stats(L)[0] == s0[0]
        # a refused key: the reference's code from either path, nothing written
        for mode in (0, 1):
            L.bee2hip_path_policy(mode)
            assert eng.bignPubkeyCalc(P, bytes(no))[0] == 504 and eng.bignSign2(P, oid, h, bytes(no), None)[0] == 504
        L.bee2hip_path_policy(0)


def test_device_fault_is_retried_then_finished_on_the_host(orc):
    eng = exp_engine()          # fault injection is a hook of the experiments build (bee2hip_internal_tune 5)
    L = eng.lib
    L.bee2hip_path_policy(0)
    H = orc.beltH()
    msg = orc.fill(1 << 16, 5)
    cs = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(cs, H[128:160], ctypes.c_size_t(32), H[192:208])
    want = orc.ctr(msg + msg, H[128:160], H[192:208])
    s0 = _stats(L)
    # one failure: the retry succeeds on the GPU
    L.bee2hip_internal_tune(5, 1)
    b = ctypes.create_string_buffer(msg, len(msg))
    L.beltCTRStepE(b, ctypes.c_size_t(len(msg)), cs)
    assert b.raw == want[: len(msg)]
    s1 = _stats(L)
    assert s1[1] == s0[1] + 1 and s1[2] == s0[2]
    # two failures: finished on the host, same bytes, same state afterwards
    L.bee2hip_internal_tune(5, 2)
    b = ctypes.create_string_buffer(msg, len(msg))
    L.beltCTRStepE(b, ctypes.c_size_t(len(msg)), cs)
    assert b.raw == want[len(msg):]
    s2 = _stats(L)
    assert s2[2] == s1[2] + 1
    L.bee2hip_internal_tune(5, 0)


def test_forced_gpu_mode_stays_fatal_on_a_device_fault():
    code = ("import ctypes, sys; sys.path[:0] = [%r, %r]; import bee2_amd; L = bee2_amd.load_experiments().lib; "
            "L.bee2hip_internal_tune(5, 2); b = ctypes.create_string_buffer(192); L.bashF(b, None); print('survived')"
            % (ROOT, os.path.join(ROOT, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BEE2HIP_FORCE="gpu"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "survived" not in r.stdout and "bashF failed" in r.stderr


@pytest.mark.parametrize("mode", ["gpu", "cpu", None])
def test_reference_test_suite_passes_in_every_mode(mode):
    """bee2's own tests (oracle/_ref/testbee2_hip, tests/test_gpu_reftests.py) with the drop-in layer forced either way"""
    binp = os.path.join(ROOT, "oracle", "_ref", "testbee2_hip")
    if not os.path.exists(binp):
        pytest.skip("oracle/_ref/testbee2_hip not built")
    env = dict(os.environ)
    env.pop("BEE2HIP_FORCE", None)
    if mode:
        env["BEE2HIP_FORCE"] = mode
    r = subprocess.run([binp], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and "Err" not in r.stdout, r.stdout + r.stderr[-1000:]
    assert r.stdout.count(": OK") == 6


def test_duplex_pipeline_of_large_host_batches_bashF(orc):
    """>= 48 MiB through bee2hip_bashF_batch: chunks uploaded by the caller's thread and downloaded by a helper thread on a
    second stream (staging.hpp duplex_inplace); ragged last chunk; every state against the oracle"""
    eng = engine()
    n = (1 << 18) + 4099                       # 51 MB, 17 chunks of 2^14 states, the last one ragged
    data = np.frombuffer(orc.fill(192 * n, 0xD0B1), dtype=np.uint8).copy()
    want = orc.bashF_batch(data.tobytes(), 8)
    assert eng.lib.bee2hip_bashF_batch(ctypes.c_void_p(data.ctypes.data), ctypes.c_size_t(n)) == 0
    assert data.tobytes() == want


def test_duplex_pipeline_of_a_large_ctr_call_keeps_the_streaming_state(orc):
    eng = engine()
    L = eng.lib
    H = orc.beltH()
    n = (100 << 20) + 5                        # 24 chunks of 4 MiB through the pipeline, 4 MiB + 5 bytes on the plain path
    kw, c0 = orc.ctr_start(H[128:160], H[192:208])
    msg = np.frombuffer(orc.fill(n + 27, 0xD0B2), dtype=np.uint8).copy()
    want = msg.copy()
    whole = (n + 27 + 15) // 16 * 16
    pad = np.zeros(whole, dtype=np.uint8)
    pad[: n + 27] = msg
    orc.ctr_blocks_np(pad, kw, c0, first=0, nthreads=8)
    want = pad[: n + 27].tobytes()
    st = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(st, H[128:160], ctypes.c_size_t(32), H[192:208])
    L.beltCTRStepE(ctypes.c_void_p(msg.ctypes.data), ctypes.c_size_t(n), st)
    L.beltCTRStepE(ctypes.c_void_p(msg.ctypes.data + n), ctypes.c_size_t(27), st)      # continues inside the last gamma block
    assert msg.tobytes() == want
    # and the one-shot form
    msg2 = np.frombuffer(orc.fill(n, 0xD0B3), dtype=np.uint8).copy()
    pad2 = np.zeros((n + 15) // 16 * 16, dtype=np.uint8)
    pad2[:n] = msg2
    orc.ctr_blocks_np(pad2, kw, c0, first=0, nthreads=8)
    out = np.empty(n, dtype=np.uint8)
    assert L.beltCTR(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(msg2.ctypes.data), ctypes.c_size_t(n), H[128:160],
                     ctypes.c_size_t(32), H[192:208]) == 0
    assert out.tobytes() == pad2[:n].tobytes()


@pytest.mark.parametrize("times", [1, 2])
def test_device_fault_in_mid_duplex_pipeline_never_encrypts_a_block_twice(orc, times):
    """the duplex CTR path downloads finished chunks into the caller's buffer while later chunks are still in
    flight.  A fault in mid-pipeline (injected: experiments build, tune 14 / 15) must leave the call's result correct: the
    retry (times = 1) or, after a second fault, the host fallback (times = 2) goes on BEHIND the chunks that came back,
    from the advanced counter -- a second pass over them would turn them into plaintext again."""
    eng = exp_engine()
    L = eng.lib
    L.bee2hip_path_policy(0)
    H = orc.beltH()
    n = (160 << 20) + 11                         # ten 16 MiB chunks; the fault hits chunk 4 of each attempt
    kw, c0 = orc.ctr_start(H[128:160], H[192:208])
    msg = np.frombuffer(orc.fill(n, 0xFA17), dtype=np.uint8).copy()
    pad = np.zeros((n + 15) // 16 * 16, dtype=np.uint8)
    pad[:n] = msg
    orc.ctr_blocks_np(pad, kw, c0, first=0, nthreads=8)
    want = pad[:n].tobytes()
    st = ctypes.create_string_buffer(L.beltCTR_keep())
    L.beltCTRStart(st, H[128:160], ctypes.c_size_t(32), H[192:208])
    s0 = _stats(L)
    L.bee2hip_internal_tune(14, 4)
    L.bee2hip_internal_tune(15, times)
    try:
        L.beltCTRStepE(ctypes.c_void_p(msg.ctypes.data), ctypes.c_size_t(n), st)
    finally:
        L.bee2hip_internal_tune(15, 0)
    s1 = _stats(L)
    assert msg.tobytes() == want
    if times == 1:
        assert s1[1] == s0[1] + 1 and s1[2] == s0[2]          # the retry finished on the GPU
    else:
        assert s1[2] == s0[2] + 1                              # two faults: the rest on the host
    # the streaming state is where a single clean pass would have left it: the next bytes continue the gamma
    more = np.frombuffer(orc.fill(37, 0xFA18), dtype=np.uint8).copy()
    pad2 = np.zeros((n + 37 + 15) // 16 * 16, dtype=np.uint8)
    pad2[:n] = np.frombuffer(orc.fill(n, 0xFA17), dtype=np.uint8)
    pad2[n: n + 37] = more
    orc.ctr_blocks_np(pad2, kw, c0, first=0, nthreads=8)
    L.beltCTRStepE(ctypes.c_void_p(more.ctypes.data), ctypes.c_size_t(37), st)
    assert more.tobytes() == pad2[n: n + 37].tobytes()
