"""ctypes binding of libbee2hip.so (include/bee2hip.h) and a small host-side mirror of
the bee2 interfaces it replaces.  Names follow bee2: bashF, bashHash, beltCTR, beltMAC,
bign128Verify ... (include/bee2/crypto/{bash,belt,bign,bign128}.h)."""
import ctypes
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_PATH = os.path.join(PKG, "lib", "libbee2hip.so")
# the same sources with -DBEE2HIP_EXPERIMENTS: rejected kernel variants kept for the A/B record and the bee2hip_internal_* /
# bee2hip_debug_fe* / bee2hip_time_kernel hooks of include/bee2hip_internal.h.  tests/ and tools/ load it where they need a
# hook; nothing a bee2 caller links against is in it only.
EXP_LIB_PATH = os.path.join(PKG, "lib", "libbee2hip_exp.so")

ERR_OK = 0
ERR_BAD_INPUT = 109
ERR_OUTOFMEMORY = 110
ERR_NOT_IMPLEMENTED = 119
ERR_BAD_OID = 301
ERR_BAD_PARAMS = 502
ERR_BAD_PUBKEY = 505
ERR_BAD_SIG = 510
ERR_BEE2HIP_DEVICE = 0x4850

OID_BELT_HASH_DER = bytes.fromhex("06092A7000020022651F51")   # bign128.c:151-153
OID_BASH384_DER = bytes.fromhex("06092A7000020022654D0C")     # bign192.c:151-153
OID_BASH512_DER = bytes.fromhex("06092A7000020022654D0D")     # bign256.c:151-153
LEVEL_OID = {128: OID_BELT_HASH_DER, 192: OID_BASH384_DER, 256: OID_BASH512_DER}
CURVE_NAME = {128: "1.2.112.0.2.0.34.101.45.3.1", 192: "1.2.112.0.2.0.34.101.45.3.2",
              256: "1.2.112.0.2.0.34.101.45.3.3"}

_sz = ctypes.c_size_t
_vp = ctypes.c_void_p
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64


class EngineError(RuntimeError):
    pass


def build(verbose=False, experiments=True):
    """compile libbee2hip.so (and, for the tests' hooks, libbee2hip_exp.so) for gfx950 in-tree, side by side
    (hipcc cross-compiles without a GPU)"""
    cmd = ["make", "-j2", "-C", os.path.join(PKG, "csrc"), "all"] + (["exp"] if experiments else [])
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


# every symbol include/bee2hip.h declares; tests check the built library exports them all
DROPIN_SYMBOLS = [
    "bashF", "bashF_deep", "bash_platform", "bashHash_keep", "bashHashStart", "bashHashStepH",
    "bashHashStepG", "bashHashStepV", "bashHash",
    "beltH", "beltKeyExpand2", "beltBlockEncr", "beltBlockEncr2", "beltBlockEncr3",
    "beltCTR_keep", "beltCTRStart", "beltCTRStepE", "beltCTR",
    "beltBlockDecr", "beltBlockDecr2", "beltBlockDecr3",
    "beltECB_keep", "beltECBStart", "beltECBStepE", "beltECBStepD", "beltECBEncr", "beltECBDecr",
    "beltCBC_keep", "beltCBCStart", "beltCBCStepE", "beltCBCStepD", "beltCBCEncr", "beltCBCDecr",
    "beltBDE_keep", "beltBDEStart", "beltBDEStepE", "beltBDEStepD", "beltBDEEncr", "beltBDEDecr",
    "beltSDE_keep", "beltSDEStart", "beltSDEStepE", "beltSDEStepD", "beltSDEEncr", "beltSDEDecr",
    "beltHash_keep", "beltHashStart", "beltHashStepH", "beltHashStepG", "beltHashStepG2", "beltHashStepV",
    "beltHashStepV2", "beltHash",
    "beltDWP_keep", "beltDWPStart", "beltDWPStepE", "beltDWPStepI", "beltDWPStepA", "beltDWPStepD", "beltDWPStepG",
    "beltDWPStepV", "beltDWPWrap", "beltDWPUnwrap",
    "beltCHE_keep", "beltCHEStart", "beltCHEStepE", "beltCHEStepI", "beltCHEStepA", "beltCHEStepD", "beltCHEStepG",
    "beltCHEStepV", "beltCHEWrap", "beltCHEUnwrap",
    "beltMAC_keep", "beltMACStart", "beltMACStepA", "beltMACStepG", "beltMACStepG2",
    "beltMACStepV", "beltMACStepV2", "beltMAC",
    "bignParamsStd", "bignVerify", "bign128Verify", "bign192Verify", "bign256Verify",
    "bignPubkeyVal", "bign128PubkeyVal", "bign192PubkeyVal", "bign256PubkeyVal",
    "bignKeypairGen", "bignPubkeyCalc", "bignSign", "bignSign2",
    "bign128KeypairGen", "bign128PubkeyCalc", "bign128Sign", "bign128Sign2",
    "bign192KeypairGen", "bign192PubkeyCalc", "bign192Sign", "bign192Sign2",
    "bign256KeypairGen", "bign256PubkeyCalc", "bign256Sign", "bign256Sign2",
]
BATCH_SYMBOLS = [
    "bee2hip_bashF_batch", "bee2hip_beltCTR_bulk", "bee2hip_bignVerify_batch",
    "bee2hip_bashHash_beltMAC_batch", "bee2hip_hash_ragged", "bee2hip_hash_ragged_dev", "bee2hip_hash_ragged_ordered_dev",
    "bee2hip_bashF_batch_dev", "bee2hip_beltCTR_blocks_dev", "bee2hip_beltBlockEncr_dev",
    "bee2hip_beltModes_blocks_dev", "bee2hip_beltCBCEncr_batch_dev", "bee2hip_beltBDE_blocks_dev", "bee2hip_beltSDE_sectors_dev", "bee2hip_beltDWP_absorb_dev", "bee2hip_beltCHE_blocks_dev",
    "bee2hip_bign128Verify_batch_dev", "bee2hip_bignVerify_batch_dev", "bee2hip_bignVerifyL_batch_dev",
    "bee2hip_bignVerify_onekey_batch", "bee2hip_bignVerifyL_onekey_batch_dev",
    "bee2hip_bignVerify_keyed_batch", "bee2hip_bignVerifyL_keyed_batch_dev",
    "bee2hip_bignVerify_onekey_batch_multi", "bee2hip_bignVerify_keyed_batch_multi",
    "bee2hip_bignVerifyL_onekey_batch_multi_dev", "bee2hip_bignVerifyL_keyed_batch_multi_dev",
    "bee2hip_bignPubkeyVal_batch", "bee2hip_bignPubkeyValL_batch_dev",
    "bee2hip_bignPubkeyCalc_batch", "bee2hip_bignSign2_batch", "bee2hip_bignSignK_batch",
    "bee2hip_device_count", "bee2hip_multi_plan", "bee2hip_bashF_batch_multi", "bee2hip_beltCTR_bulk_multi",
    "bee2hip_bignVerify_batch_multi", "bee2hip_bignSign2_batch_multi", "bee2hip_bashHash_beltMAC_batch_multi",
    "bee2hip_hash_ragged_multi",
    "bee2hip_bashF_batch_multi_dev", "bee2hip_beltCTR_blocks_multi_dev", "bee2hip_bignVerifyL_batch_multi_dev",
    "bee2hip_bashHash_beltMAC_batch_multi_dev",
    "bee2hip_bignPubkeyCalcL_batch_dev", "bee2hip_bignSign2L_batch_dev", "bee2hip_bignSignKL_batch_dev",
    "bee2hip_bashHash_beltMAC_batch_dev",
    "bee2hip_set_device", "bee2hip_sync", "bee2hip_last_error", "bee2hip_version", "bee2hip_path_policy", "bee2hip_path_count",
]
# include/bee2hip_internal.h: test / bench hooks, not product ABI
INTERNAL_SYMBOLS = ["bee2hip_time_kernel", "bee2hip_debug_fe", "bee2hip_debug_feL", "bee2hip_internal_tune",
                    "bee2hip_internal_clock_probe", "bee2hip_internal_stat"]


def lib_exports(path=LIB_PATH):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


class bign_params(ctypes.Structure):            # include/bee2/crypto/bign.h:65-74
    _fields_ = [("l", _sz), ("p", ctypes.c_ubyte * 64), ("a", ctypes.c_ubyte * 64),
                ("b", ctypes.c_ubyte * 64), ("q", ctypes.c_ubyte * 64),
                ("yG", ctypes.c_ubyte * 64), ("seed", ctypes.c_ubyte * 8)]


class Engine:
    """One per process/GPU.  Device-pointer methods take torch CUDA tensors (uint8/int32,
    contiguous) and are asynchronous on torch's current stream."""

    def __init__(self, lib):
        self.lib = L = lib
        for name in ("bee2hip_last_error", "bee2hip_version"):
            getattr(L, name).restype = ctypes.c_char_p
        L.beltH.restype = ctypes.POINTER(ctypes.c_ubyte)
        for name in ("bee2hip_internal_stat", "bee2hip_path_count"):
            if hasattr(L, name):
                getattr(L, name).restype = ctypes.c_ulonglong
        if hasattr(L, "bee2hip_path_policy"):
            L.bee2hip_path_policy.restype = ctypes.c_int
        self.experiments = hasattr(L, "bee2hip_internal_tune")
        for name in ("bashF_deep", "bashHash_keep", "beltCTR_keep", "beltMAC_keep", "beltECB_keep", "beltCBC_keep"):
            if hasattr(L, name):
                getattr(L, name).restype = _sz
        for name in DROPIN_SYMBOLS + BATCH_SYMBOLS + INTERNAL_SYMBOLS:
            f = getattr(L, name, None)
            if f is not None and name.startswith(("bee2hip_", "bash", "belt", "bign")) and \
                    name not in ("bee2hip_last_error", "bee2hip_version", "beltH", "bashF_deep",
                                 "bashHash_keep", "beltCTR_keep", "beltMAC_keep", "beltECB_keep",
                                 "beltCBC_keep", "bash_platform", "bee2hip_device_count", "bee2hip_internal_stat",
                                 "bee2hip_path_count", "bee2hip_path_policy"):
                f.restype = _u32

    # ------------------------------------------------------------------ util
    def _check(self, code, what):
        if code != ERR_OK:
            raise EngineError(f"{what}: err {code} {self.lib.bee2hip_last_error().decode()}")

    def version(self):
        return self.lib.bee2hip_version().decode()

    def set_device(self, dev):
        self._check(self.lib.bee2hip_set_device(int(dev)), "bee2hip_set_device")

    @staticmethod
    def _stream():
        import torch
        return _vp(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _ptr(t):
        assert t.is_cuda and t.is_contiguous()
        return _vp(t.data_ptr())

    def sync(self):
        self._check(self.lib.bee2hip_sync(self._stream()), "bee2hip_sync")

    # --------------------------------------------------- device-pointer batch
    def bashF_batch_dev(self, states):
        """states: uint8 CUDA tensor of n*192 bytes, permuted in place"""
        n = states.numel() // 192
        assert states.numel() == n * 192
        self._check(self.lib.bee2hip_bashF_batch_dev(self._ptr(states), _sz(n), self._stream()),
                    "bashF_batch_dev")

    def beltCTR_blocks_dev(self, buf, key_words, ctr0_words, first_block=0):
        """buf: uint8 CUDA tensor, multiple of 16 bytes; key_words 32 B, ctr0_words 16 B
        (host bytes holding u32[8] / u32[4] little-endian)"""
        n = buf.numel() // 16
        assert buf.numel() == n * 16
        self._check(self.lib.bee2hip_beltCTR_blocks_dev(self._ptr(buf), _sz(n), bytes(key_words),
                                                        bytes(ctr0_words), _u64(first_block),
                                                        self._stream()), "beltCTR_blocks_dev")

    def beltBlockEncr_dev(self, blocks, key_words):
        n = blocks.numel() // 16
        self._check(self.lib.bee2hip_beltBlockEncr_dev(self._ptr(blocks), _sz(n), bytes(key_words),
                                                       self._stream()), "beltBlockEncr_dev")

    def beltModes_blocks_dev(self, mode, src, dst, key_words, iv_words=None):
        """mode 0 ECB encrypt, 1 ECB decrypt, 2 CBC decrypt (src != dst); full blocks"""
        n = src.numel() // 16
        assert src.numel() == 16 * n and dst.numel() == 16 * n
        self._check(self.lib.bee2hip_beltModes_blocks_dev(int(mode), self._ptr(src), self._ptr(dst), _sz(n),
                                                          bytes(key_words), bytes(iv_words) if iv_words else None,
                                                          self._stream()), "beltModes_blocks_dev")

    # ---- belt-dwp (8f-2)
    def dwp_steps(self, key, iv, ops, mode="DWP"):
        """ops = [("E"|"D"|"I"|"A", bytes) | ("G",) | ("V", mac)] applied to one belt-dwp (mode "DWP") or
        belt-che ("CHE") state; -> (E/D outputs concatenated, [mac per "G"], [bool per "V"])"""
        f = lambda name: getattr(self.lib, f"belt{mode}{name}")
        st = ctypes.create_string_buffer(f("_keep")())
        f("Start")(st, bytes(key), _sz(len(key)), bytes(iv))
        out, macs, oks = b"", [], []
        for op in ops:
            if op[0] in "ED":
                b = ctypes.create_string_buffer(bytes(op[1]), max(len(op[1]), 1))
                f("Step" + op[0])(b, _sz(len(op[1])), st)
                out += b.raw[: len(op[1])]
            elif op[0] in "IA":
                f("Step" + op[0])(bytes(op[1]), _sz(len(op[1])), st)
            elif op[0] == "G":
                m = ctypes.create_string_buffer(8)
                f("StepG")(m, st)
                macs.append(m.raw)
            else:
                oks.append(bool(f("StepV")(bytes(op[1]), st)))
        return out, macs, oks

    def dwp_wrap(self, crit, open_, key, iv, mode="DWP"):
        dest, mac = ctypes.create_string_buffer(max(len(crit), 1)), ctypes.create_string_buffer(8)
        code = getattr(self.lib, f"belt{mode}Wrap")(dest, mac, bytes(crit), _sz(len(crit)), bytes(open_), _sz(len(open_)),
                                    bytes(key), _sz(len(key)), bytes(iv))
        return code, dest.raw[: len(crit)], mac.raw

    def dwp_unwrap(self, crit, open_, mac, key, iv, mode="DWP"):
        dest = ctypes.create_string_buffer(max(len(crit), 1))
        code = getattr(self.lib, f"belt{mode}Unwrap")(dest, bytes(crit), _sz(len(crit)), bytes(open_), _sz(len(open_)),
                                      bytes(mac), bytes(key), _sz(len(key)), bytes(iv))
        return code, dest.raw[: len(crit)]

    def beltDWPStart(self, key, iv):
        """-> (expanded key, ctr0 = E_K(iv), r = E_K(ctr0), t0 = H[0..16)) as bytes of u32 words"""
        st = ctypes.create_string_buffer(self.lib.beltDWP_keep())
        self.lib.beltDWPStart(st, bytes(key), _sz(len(key)), bytes(iv))
        n = self.lib.beltCTR_keep()
        return st.raw[:32], st.raw[32:48], st.raw[n:n + 16], st.raw[n + 16:n + 32]

    def beltCHEStart(self, key, iv):
        """-> (expanded key, s0 = r = E_K(iv), t0 = H[0..16)) as bytes of u32 words"""
        st = ctypes.create_string_buffer(self.lib.beltCHE_keep())
        self.lib.beltCHEStart(st, bytes(key), _sz(len(key)), bytes(iv))
        n = self.lib.beltCTR_keep()
        return st.raw[:32], st.raw[n:n + 16], st.raw[n + 16:n + 32]

    def beltCHE_blocks_dev(self, src, dst, key_words, s_words, first_block=0, s_out=None):
        """belt-che keystream over whole blocks in HBM; the piece starts first_block blocks into the stream"""
        n = src.numel() // 16
        assert src.numel() == 16 * n and dst.numel() == 16 * n
        self._check(self.lib.bee2hip_beltCHE_blocks_dev(self._ptr(src), self._ptr(dst), _sz(n), bytes(key_words),
                                                        bytes(s_words), ctypes.c_uint64(first_block),
                                                        self._ptr(s_out) if s_out is not None else None,
                                                        self._stream()), "beltCHE_blocks_dev")

    def beltDWP_absorb_dev(self, data, nbytes, r_words, t_words, t_out):
        """t_out (16-byte device tensor) <- t after absorbing nbytes of the device tensor `data`"""
        self._check(self.lib.bee2hip_beltDWP_absorb_dev(self._ptr(data) if nbytes else None, _sz(nbytes),
                                                        bytes(r_words), bytes(t_words), self._ptr(t_out),
                                                        self._stream()), "beltDWP_absorb_dev")

    def beltSDE_sectors_dev(self, decr, sectors, sector_bytes, key_words, ivs):
        """belt-sde over contiguous sectors in HBM, in place; ivs = n x 16-byte device tensor"""
        n = ivs.numel() // 16
        assert sectors.numel() == n * sector_bytes
        self._check(self.lib.bee2hip_beltSDE_sectors_dev(int(decr), self._ptr(sectors), _sz(sector_bytes), _sz(n),
                                                         bytes(key_words), self._ptr(ivs), self._stream()),
                    "beltSDE_sectors_dev")

    def beltBDEStart(self, key, iv):
        """-> (expanded key, s = E_K(iv)) as 32 + 16 bytes of u32 words, for beltBDE_blocks_dev"""
        st = ctypes.create_string_buffer(self.lib.beltBDE_keep())
        self.lib.beltBDEStart(st, bytes(key), _sz(len(key)), bytes(iv))
        return st.raw[:32], st.raw[32:48]

    def beltBDE_blocks_dev(self, decr, src, dst, key_words, s_words, first_block=0, s_out=None):
        """belt-bde on whole blocks in HBM; the piece starts first_block blocks into the stream;
        s_out (16-byte device tensor, optional) receives the state after the piece"""
        n = src.numel() // 16
        assert src.numel() == 16 * n and dst.numel() == 16 * n
        self._check(self.lib.bee2hip_beltBDE_blocks_dev(int(decr), self._ptr(src), self._ptr(dst), _sz(n),
                                                        bytes(key_words), bytes(s_words),
                                                        ctypes.c_uint64(first_block),
                                                        self._ptr(s_out) if s_out is not None else None,
                                                        self._stream()), "beltBDE_blocks_dev")

    def beltCBCEncr_batch_dev(self, msgs, nblk, key_words, ivs):
        n = ivs.numel() // 16
        assert msgs.numel() == n * nblk * 16
        self._check(self.lib.bee2hip_beltCBCEncr_batch_dev(self._ptr(msgs), _sz(nblk), _sz(n), bytes(key_words),
                                                           self._ptr(ivs), self._stream()), "beltCBCEncr_batch_dev")

    def bign128Verify_batch_dev(self, hashes, sigs, pubkeys, codes):
        n = hashes.numel() // 32
        assert sigs.numel() == 48 * n and pubkeys.numel() == 64 * n and codes.numel() >= n
        self._check(self.lib.bee2hip_bign128Verify_batch_dev(
            self._ptr(hashes), self._ptr(sigs), self._ptr(pubkeys), _sz(n), self._ptr(codes),
            self._stream()), "bign128Verify_batch_dev")

    def bignVerifyL_batch_dev(self, l, oid_der, hashes, sigs, pubkeys, codes):
        n = hashes.numel() // (l // 4)
        assert sigs.numel() == (3 * l // 8) * n and pubkeys.numel() == (l // 2) * n and codes.numel() >= n
        self._check(self.lib.bee2hip_bignVerifyL_batch_dev(
            _sz(l), bytes(oid_der), _sz(len(oid_der)), self._ptr(hashes), self._ptr(sigs), self._ptr(pubkeys),
            _sz(n), self._ptr(codes), self._stream()), "bignVerifyL_batch_dev")

    def bignVerifyL_onekey_batch_dev(self, l, oid_der, hashes, sigs, pubkey, codes):
        """n signatures under ONE public key (host bytes, l/2 octets); hashes / sigs / codes device tensors"""
        n = hashes.numel() // (l // 4)
        assert sigs.numel() == (3 * l // 8) * n and len(pubkey) == l // 2 and codes.numel() >= n
        self._check(self.lib.bee2hip_bignVerifyL_onekey_batch_dev(
            _sz(l), bytes(oid_der), _sz(len(oid_der)), self._ptr(hashes), self._ptr(sigs), bytes(pubkey),
            _sz(n), self._ptr(codes), self._stream()), "bignVerifyL_onekey_batch_dev")

    def bignVerifyL_keyed_batch_dev(self, l, oid_der, hashes, sigs, pubkeys, key_index, codes):
        """n signatures of K signers: pubkeys = K keys (host bytes), key_index = n x int32 / uint32 device tensor"""
        n = hashes.numel() // (l // 4)
        nkeys = len(pubkeys) // (l // 2)
        assert sigs.numel() == (3 * l // 8) * n and key_index.numel() == n and codes.numel() >= n and key_index.element_size() == 4
        self._check(self.lib.bee2hip_bignVerifyL_keyed_batch_dev(
            _sz(l), bytes(oid_der), _sz(len(oid_der)), self._ptr(hashes), self._ptr(sigs), bytes(pubkeys), _sz(nkeys),
            self._ptr(key_index), _sz(n), self._ptr(codes), self._stream()), "bignVerifyL_keyed_batch_dev")

    def bignPubkeyValL_batch_dev(self, l, pubkeys, codes):
        n = pubkeys.numel() // (l // 2)
        assert codes.numel() >= n
        self._check(self.lib.bee2hip_bignPubkeyValL_batch_dev(_sz(l), self._ptr(pubkeys), _sz(n), self._ptr(codes),
                                                              self._stream()), "bignPubkeyValL_batch_dev")

    def bashHash_beltMAC_batch_dev(self, msgs, msg_len, l, key, digests, tags, n=None):
        if n is None:
            n = msgs.numel() // msg_len if msg_len else 0
        self._check(self.lib.bee2hip_bashHash_beltMAC_batch_dev(
            self._ptr(msgs), _sz(msg_len), _sz(n), _sz(l), bytes(key), _sz(len(key)),
            self._ptr(digests) if digests is not None else None,
            self._ptr(tags) if tags is not None else None, self._stream()),
            "bashHash_beltMAC_batch_dev")

    def hash_ragged_dev(self, alg, data, offsets, digests, n, order=None):
        """alg 0 = belt-hash, 128/192/256 = bash256/384/512; data (u8), offsets (int64, n+1),
        digests (u8) and the optional launch order (int32 permutation, longest message first) are
        device tensors"""
        o = self._ptr(order) if order is not None else None
        self._check(self.lib.bee2hip_hash_ragged_ordered_dev(_sz(alg), self._ptr(data), self._ptr(offsets), o,
                                                             _sz(n), self._ptr(digests), self._stream()),
                    "hash_ragged_ordered_dev")

    def time_kernel(self, which, reps, a=None, b=None, c=None, d=None, n=0, aux=0):
        ms = ctypes.c_float(0)
        p = [self._ptr(t) if t is not None else None for t in (a, b, c, d)]
        self._check(self.lib.bee2hip_time_kernel(int(which), int(reps), p[0], p[1], p[2], p[3],
                                                 _sz(n), _sz(aux), self._stream(),
                                                 ctypes.byref(ms)), "time_kernel")
        return ms.value

    # ------------------------------------------------------ host-pointer batch
    def bashF_batch(self, states):
        buf = ctypes.create_string_buffer(bytes(states), len(states))
        self._check(self.lib.bee2hip_bashF_batch(buf, _sz(len(states) // 192)), "bashF_batch")
        return buf.raw

    def bignVerify_batch(self, hashes, sigs, pubkeys, oid_der=OID_BELT_HASH_DER, params=None):
        if params is None:
            params = self.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
        n = len(hashes) // (params.l // 4)
        codes = (_u32 * max(n, 1))()
        code = self.lib.bee2hip_bignVerify_batch(ctypes.byref(params), bytes(oid_der),
                                                 _sz(len(oid_der)), bytes(hashes), bytes(sigs),
                                                 bytes(pubkeys), _sz(n), codes)
        return code, list(codes)[:n]

    def bignVerify_onekey_batch(self, hashes, sigs, pubkey, oid_der=OID_BELT_HASH_DER, params=None):
        if params is None:
            params = self.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
        n = len(hashes) // (params.l // 4)
        codes = (_u32 * max(n, 1))()
        code = self.lib.bee2hip_bignVerify_onekey_batch(ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)),
                                                        bytes(hashes), bytes(sigs), bytes(pubkey), _sz(n), codes)
        return code, list(codes)[:n]

    def bignVerify_keyed_batch(self, hashes, sigs, pubkeys, key_index, oid_der=OID_BELT_HASH_DER, params=None):
        if params is None:
            params = self.bignParamsStd("1.2.112.0.2.0.34.101.45.3.1")
        n = len(hashes) // (params.l // 4)
        nkeys = len(pubkeys) // (params.l // 2)
        codes = (_u32 * max(n, 1))()
        idx = (_u32 * max(n, 1))(*key_index)
        code = self.lib.bee2hip_bignVerify_keyed_batch(ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)), bytes(hashes),
                                                       bytes(sigs), bytes(pubkeys), _sz(nkeys), idx, _sz(n), codes)
        return code, list(codes)[:n]

    def bignPubkeyVal_batch(self, pubkeys, params):
        n = len(pubkeys) // (params.l // 2)
        codes = (_u32 * max(n, 1))()
        code = self.lib.bee2hip_bignPubkeyVal_batch(ctypes.byref(params), bytes(pubkeys), _sz(n), codes)
        return code, list(codes)[:n]

    def bashHash_beltMAC_batch(self, msgs, msg_len, l, key, want_hash=True, want_mac=True, n=None):
        if n is None:
            n = len(msgs) // msg_len if msg_len else 0
        dig = ctypes.create_string_buffer(max(1, n * (l // 4))) if want_hash else None
        tag = ctypes.create_string_buffer(max(1, n * 8)) if want_mac else None
        self._check(self.lib.bee2hip_bashHash_beltMAC_batch(bytes(msgs), _sz(msg_len), _sz(n), _sz(l),
                                                            bytes(key), _sz(len(key)), dig, tag),
                    "bashHash_beltMAC_batch")
        return (dig.raw[: n * (l // 4)] if dig else None), (tag.raw[: n * 8] if tag else None)

    def hash_ragged(self, alg, messages):
        """alg 0 = belt-hash, 128/192/256 = bash256/384/512; messages: list of bytes"""
        import struct
        offs = [0]
        for m in messages:
            offs.append(offs[-1] + len(m))
        data = b"".join(messages)
        n = len(messages)
        dlen = alg // 4 if alg else 32
        out = ctypes.create_string_buffer(max(1, n * dlen))
        code = self.lib.bee2hip_hash_ragged(_sz(alg), data, struct.pack(f"<{n + 1}Q", *offs), _sz(n), out)
        return code, [out.raw[i * dlen:(i + 1) * dlen] for i in range(n)]

    # ------------------------------------------------- bee2 drop-in interface
    def beltH(self):
        p = self.lib.beltH()
        return bytes(p[i] for i in range(256))

    def bashF(self, block):
        b = ctypes.create_string_buffer(bytes(block), 192)
        self.lib.bashF(b, None)
        return b.raw

    def bashHash(self, l, src):
        out = ctypes.create_string_buffer(max(l // 4, 1))
        code = self.lib.bashHash(out, _sz(l), bytes(src), _sz(len(src)))
        return code, out.raw[: l // 4]

    def bashHash_steps(self, l, src, splits):
        st = ctypes.create_string_buffer(self.lib.bashHash_keep())
        self.lib.bashHashStart(st, _sz(l))
        off = 0
        for s in splits:
            self.lib.bashHashStepH(bytes(src[off:off + s]), _sz(s), st)
            off += s
        out = ctypes.create_string_buffer(l // 4)
        self.lib.bashHashStepG(out, _sz(l // 4), st)
        ok = self.lib.bashHashStepV(out, _sz(l // 4), st)
        return out.raw, bool(ok)

    def beltKeyExpand2(self, key):
        k = (_u32 * 8)()
        self.lib.beltKeyExpand2(k, bytes(key), _sz(len(key)))
        return bytes(k)

    def beltBlockEncr(self, block, key):
        b = ctypes.create_string_buffer(bytes(block), 16)
        self.lib.beltBlockEncr(b, self.beltKeyExpand2(key))
        return b.raw

    def beltCTR(self, src, key, iv):
        out = ctypes.create_string_buffer(max(len(src), 1))
        code = self.lib.beltCTR(out, bytes(src), _sz(len(src)), bytes(key), _sz(len(key)), bytes(iv))
        return code, out.raw[: len(src)]

    def beltCTR_steps(self, src, key, iv, splits):
        """Start / StepE*; returns (ciphertext, final state bytes)"""
        st = ctypes.create_string_buffer(self.lib.beltCTR_keep())
        self.lib.beltCTRStart(st, bytes(key), _sz(len(key)), bytes(iv))
        buf = ctypes.create_string_buffer(bytes(src), max(len(src), 1))
        off = 0
        for s in splits:
            self.lib.beltCTRStepE(ctypes.byref(buf, off), _sz(s), st)
            off += s
        return buf.raw[: len(src)], st.raw

    def beltCTRStart(self, key, iv):
        st = ctypes.create_string_buffer(self.lib.beltCTR_keep())
        self.lib.beltCTRStart(st, bytes(key), _sz(len(key)), bytes(iv))
        return st.raw[:32], st.raw[32:48]

    def beltBlockDecr(self, block, key):
        b = ctypes.create_string_buffer(bytes(block), 16)
        self.lib.beltBlockDecr(b, self.beltKeyExpand2(key))
        return b.raw

    def belt_mode(self, fn, src, key, iv=None):
        """one-shot beltECBEncr / beltECBDecr / beltCBCEncr / beltCBCDecr / beltBDEEncr / beltBDEDecr"""
        out = ctypes.create_string_buffer(max(len(src), 1))
        f = getattr(self.lib, fn)
        if iv is None:
            code = f(out, bytes(src), _sz(len(src)), bytes(key), _sz(len(key)))
        else:
            code = f(out, bytes(src), _sz(len(src)), bytes(key), _sz(len(key)), bytes(iv))
        return code, out.raw[: len(src)]

    def belt_mode_steps(self, mode, decr, src, key, iv, splits):
        """Start / Step{E,D}* of beltECB (mode 'ECB'), beltCBC ('CBC'), beltBDE ('BDE'); for 'SDE' every split
        is one sector with the same iv (beltSDEStep{E,D}(buf, count, iv, state), belt_sde.c:47-71)"""
        st = ctypes.create_string_buffer(getattr(self.lib, f"belt{mode}_keep")())
        if mode in ("ECB", "SDE"):
            getattr(self.lib, f"belt{mode}Start")(st, bytes(key), _sz(len(key)))
        else:
            getattr(self.lib, f"belt{mode}Start")(st, bytes(key), _sz(len(key)), bytes(iv))
        step = getattr(self.lib, f"belt{mode}Step{'D' if decr else 'E'}")
        buf = ctypes.create_string_buffer(bytes(src), len(src))
        off = 0
        for s in splits:
            if mode == "SDE":
                step(ctypes.byref(buf, off), _sz(s), bytes(iv), st)
            else:
                step(ctypes.byref(buf, off), _sz(s), st)
            off += s
        return buf.raw[: len(src)]

    def beltHash(self, src):
        out = ctypes.create_string_buffer(32)
        code = self.lib.beltHash(out, bytes(src), _sz(len(src)))
        return code, out.raw

    def beltHash_steps(self, src, splits, hash_len=32):
        """Start / StepH* / StepG2 with a tag taken after every piece; -> [digest prefix after each split]"""
        st = ctypes.create_string_buffer(self.lib.beltHash_keep())
        self.lib.beltHashStart(st)
        outs, off = [], 0
        for s in splits:
            self.lib.beltHashStepH(bytes(src[off:off + s]), _sz(s), st)
            off += s
            d = ctypes.create_string_buffer(32)
            self.lib.beltHashStepG2(d, _sz(hash_len), st)
            outs.append(d.raw[:hash_len])
            assert self.lib.beltHashStepV2(d.raw[:hash_len], _sz(hash_len), st) == 1
        return outs

    def beltMAC(self, src, key):
        out = ctypes.create_string_buffer(8)
        code = self.lib.beltMAC(out, bytes(src), _sz(len(src)), bytes(key), _sz(len(key)))
        return code, out.raw

    def beltMAC_steps(self, src, key, splits):
        st = ctypes.create_string_buffer(self.lib.beltMAC_keep())
        self.lib.beltMACStart(st, bytes(key), _sz(len(key)))
        off = 0
        for s in splits:
            self.lib.beltMACStepA(bytes(src[off:off + s]), _sz(s), st)
            off += s
        out = ctypes.create_string_buffer(8)
        self.lib.beltMACStepG(out, st)
        ok = self.lib.beltMACStepV(out, st)
        out4 = ctypes.create_string_buffer(4)
        self.lib.beltMACStepG2(out4, _sz(4), st)
        ok2 = self.lib.beltMACStepV2(out4, _sz(4), st)
        return out.raw, bool(ok) and bool(ok2) and out4.raw == out.raw[:4]

    def bignParamsStd(self, name):
        p = bign_params()
        code = self.lib.bignParamsStd(ctypes.byref(p), name.encode())
        if code != ERR_OK:
            raise EngineError(f"bignParamsStd({name}): err {code}")
        return p

    def bignVerify(self, params, oid_der, hash_, sig, pubkey):
        return self.lib.bignVerify(ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)),
                                   bytes(hash_), bytes(sig), bytes(pubkey))

    def bign128Verify(self, hash_, sig, pubkey):
        return self.lib.bign128Verify(bytes(hash_), bytes(sig), bytes(pubkey))

    def bignPubkeyVal(self, params, pubkey):
        return self.lib.bignPubkeyVal(ctypes.byref(params), bytes(pubkey))

    def bignLPubkeyVal(self, l, pubkey):
        """bign128PubkeyVal / bign192PubkeyVal / bign256PubkeyVal"""
        return getattr(self.lib, f"bign{l}PubkeyVal")(bytes(pubkey))

    def bignLVerify(self, l, hash_, sig, pubkey):
        """bign128Verify / bign192Verify / bign256Verify"""
        return getattr(self.lib, f"bign{l}Verify")(bytes(hash_), bytes(sig), bytes(pubkey))

    # ------------------------------------- 8f-4 tail: keygen / pubkey calc / sign
    GEN_I = ctypes.CFUNCTYPE(None, ctypes.c_void_p, _sz, ctypes.c_void_p)

    @staticmethod
    def rng_from_bytes(stream):
        """gen_i callback that hands out consecutive bytes of `stream` (bytes); .used tells how many went"""
        pos = [0]

        def gen(buf, count, _state):
            chunk = stream[pos[0]:pos[0] + count]
            assert len(chunk) == count, "rng stream exhausted"
            ctypes.memmove(buf, chunk, count)
            pos[0] += count
        cb = Engine.GEN_I(gen)
        cb.pos = pos
        return cb

    def bignPubkeyCalc(self, params, privkey):
        out = ctypes.create_string_buffer(params.l // 2)
        code = self.lib.bignPubkeyCalc(out, ctypes.byref(params), bytes(privkey))
        return code, out.raw

    def bignLPubkeyCalc(self, l, privkey):
        out = ctypes.create_string_buffer(l // 2)
        return getattr(self.lib, f"bign{l}PubkeyCalc")(out, bytes(privkey)), out.raw

    def bignKeypairGen(self, params, rng_cb):
        priv = ctypes.create_string_buffer(params.l // 4)
        pub = ctypes.create_string_buffer(params.l // 2)
        code = self.lib.bignKeypairGen(priv, pub, ctypes.byref(params), rng_cb, None)
        return code, priv.raw, pub.raw

    def bignLKeypairGen(self, l, rng_cb):
        priv = ctypes.create_string_buffer(l // 4)
        pub = ctypes.create_string_buffer(l // 2)
        code = getattr(self.lib, f"bign{l}KeypairGen")(priv, pub, rng_cb, None)
        return code, priv.raw, pub.raw

    def bignSign2(self, params, oid_der, hash_, privkey, t=None):
        sig = ctypes.create_string_buffer(3 * params.l // 8)
        code = self.lib.bignSign2(sig, ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)), bytes(hash_),
                                  bytes(privkey), t, _sz(len(t) if t else 0))
        return code, sig.raw

    def bignLSign2(self, l, hash_, privkey, t=None):
        sig = ctypes.create_string_buffer(3 * l // 8)
        code = getattr(self.lib, f"bign{l}Sign2")(sig, bytes(hash_), bytes(privkey), t, _sz(len(t) if t else 0))
        return code, sig.raw

    def bignSign(self, params, oid_der, hash_, privkey, rng_cb):
        sig = ctypes.create_string_buffer(3 * params.l // 8)
        code = self.lib.bignSign(sig, ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)), bytes(hash_),
                                 bytes(privkey), rng_cb, None)
        return code, sig.raw

    def bignLSign(self, l, hash_, privkey, rng_cb):
        sig = ctypes.create_string_buffer(3 * l // 8)
        code = getattr(self.lib, f"bign{l}Sign")(sig, bytes(hash_), bytes(privkey), rng_cb, None)
        return code, sig.raw

    def bignPubkeyCalc_batch(self, params, privkeys):
        n = len(privkeys) // (params.l // 4)
        pubs = ctypes.create_string_buffer(max(1, n * params.l // 2))
        codes = (_u32 * max(n, 1))()
        code = self.lib.bee2hip_bignPubkeyCalc_batch(ctypes.byref(params), bytes(privkeys), _sz(n), pubs, codes)
        return code, pubs.raw[: n * params.l // 2], list(codes)[:n]

    def bignSign2_batch(self, params, oid_der, hashes, privkeys, t=None):
        n = len(hashes) // (params.l // 4)
        sg = 3 * params.l // 8
        sigs = ctypes.create_string_buffer(max(1, n * sg))
        codes = (_u32 * max(n, 1))()
        code = self.lib.bee2hip_bignSign2_batch(ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)), bytes(hashes),
                                                bytes(privkeys), t, _sz(len(t) if t else 0), _sz(n), sigs, codes)
        return code, sigs.raw[: n * sg], list(codes)[:n]

    def bignSignK_batch(self, params, oid_der, hashes, privkeys, ks):
        n = len(hashes) // (params.l // 4)
        sg = 3 * params.l // 8
        sigs = ctypes.create_string_buffer(max(1, n * sg))
        codes = (_u32 * max(n, 1))()
        code = self.lib.bee2hip_bignSignK_batch(ctypes.byref(params), bytes(oid_der), _sz(len(oid_der)), bytes(hashes),
                                                bytes(privkeys), bytes(ks), _sz(n), sigs, codes)
        return code, sigs.raw[: n * sg], list(codes)[:n]

    def bignPubkeyCalcL_batch_dev(self, l, privkeys, pubkeys, codes):
        n = privkeys.numel() // (l // 4)
        self._check(self.lib.bee2hip_bignPubkeyCalcL_batch_dev(_sz(l), self._ptr(privkeys), _sz(n), self._ptr(pubkeys),
                                                               self._ptr(codes), self._stream()), "bignPubkeyCalcL_batch_dev")

    def bignSign2L_batch_dev(self, l, oid_der, hashes, privkeys, sigs, codes, t=None, t_len=0, t_shared=True):
        n = hashes.numel() // (l // 4)
        self._check(self.lib.bee2hip_bignSign2L_batch_dev(
            _sz(l), bytes(oid_der), _sz(len(oid_der)), self._ptr(hashes), self._ptr(privkeys),
            self._ptr(t) if t is not None else None, _sz(t_len), int(bool(t_shared)), _sz(n), self._ptr(sigs),
            self._ptr(codes), self._stream()), "bignSign2L_batch_dev")

    def bignSignKL_batch_dev(self, l, oid_der, hashes, privkeys, ks, sigs, codes):
        n = hashes.numel() // (l // 4)
        self._check(self.lib.bee2hip_bignSignKL_batch_dev(
            _sz(l), bytes(oid_der), _sz(len(oid_der)), self._ptr(hashes), self._ptr(privkeys), self._ptr(ks), _sz(n),
            self._ptr(sigs), self._ptr(codes), self._stream()), "bignSignKL_batch_dev")


_engine = None


def load(path=LIB_PATH):
    """Load libbee2hip.so.  Fails loudly if it is missing -- there is no fallback path."""
    global _engine
    if _engine is None:
        path = os.environ.get("BEE2HIP_LIB", path)       # experiments only (tools/ubench)
        if not os.path.exists(path):
            raise EngineError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(bee2_amd has no CPU fallback)")
        _engine = Engine(ctypes.CDLL(path))
    return _engine


_exp_engine = None


def load_experiments(path=EXP_LIB_PATH):
    """Load libbee2hip_exp.so (the -DBEE2HIP_EXPERIMENTS build: test hooks + the A/B variants).  A second, independent copy of
    the library in the process: its own staging buffers, scratch pools and policy switches."""
    global _exp_engine
    if _exp_engine is None:
        path = os.environ.get("BEE2HIP_EXP_LIB", path)
        if not os.path.exists(path):
            raise EngineError(f"{path} not found: `make -C bee2_amd/csrc exp` (or __graft_entry__.build())")
        _exp_engine = Engine(ctypes.CDLL(path))
        assert _exp_engine.experiments, f"{path} was not built with -DBEE2HIP_EXPERIMENTS"
    return _exp_engine
