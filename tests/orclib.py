"""ctypes view of oracle/liboracle.so -- the CPU checker (test infrastructure).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_SO = os.path.join(ROOT, "oracle", "liboracle.so")
GOLD = os.path.join(ROOT, "tests", "golden")
_sz = ctypes.c_size_t
_u8p = ctypes.c_char_p


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_beltH.restype = ctypes.POINTER(ctypes.c_ubyte)
        lib.orc_bashHash.restype = ctypes.c_uint32
        lib.orc_beltCTR.restype = ctypes.c_uint32
        lib.orc_beltMAC.restype = ctypes.c_uint32
        lib.orc_beltHash.restype = ctypes.c_uint32
        lib.orc_bign128Verify.restype = ctypes.c_uint32
        lib.orc_bign128Verify_ex.restype = ctypes.c_uint32

    # --- bash
    def beltH(self):
        p = self.lib.orc_beltH()
        return bytes(p[i] for i in range(256))

    def bashF(self, state):
        b = ctypes.create_string_buffer(bytes(state), 192)
        self.lib.orc_bashF(b)
        return b.raw

    def bashF_batch(self, states, nthreads=1):
        n = len(states) // 192
        b = ctypes.create_string_buffer(bytes(states), len(states))
        self.lib.orc_bashF_batch(b, _sz(n), ctypes.c_int(nthreads))
        return b.raw

    def bashF_batch_np(self, arr, nthreads=1):
        """in-place on a contiguous numpy uint8 array of n*192 bytes"""
        assert arr.flags["C_CONTIGUOUS"] and arr.nbytes % 192 == 0
        self.lib.orc_bashF_batch(ctypes.c_void_p(arr.ctypes.data), _sz(arr.nbytes // 192),
                                 ctypes.c_int(nthreads))

    def bashHash(self, l, msg):
        out = ctypes.create_string_buffer(max(l // 4, 1))
        code = self.lib.orc_bashHash(out, _sz(l), bytes(msg), _sz(len(msg)))
        return code, out.raw[: l // 4]

    def bashHash_steps(self, l, msg, splits):
        st = ctypes.create_string_buffer(192 + 16)
        self.lib.orc_bashHashStart(st, _sz(l))
        off = 0
        for s in splits:
            self.lib.orc_bashHashStepH(bytes(msg[off:off + s]), _sz(s), st)
            off += s
        out = ctypes.create_string_buffer(l // 4)
        self.lib.orc_bashHashStepG(out, _sz(l // 4), st)
        return out.raw

    # --- belt
    def key_expand(self, key):
        k = (ctypes.c_uint32 * 8)()
        self.lib.orc_beltKeyExpand2(k, bytes(key), _sz(len(key)))
        return k

    def block_encr(self, block, key):
        b = ctypes.create_string_buffer(bytes(block), 16)
        self.lib.orc_beltBlockEncr(b, self.key_expand(key))
        return b.raw

    def ctr(self, msg, key, iv, splits=None):
        st = ctypes.create_string_buffer(32 + 16 + 16 + 8)
        self.lib.orc_beltCTRStart(st, bytes(key), _sz(len(key)), bytes(iv))
        buf = ctypes.create_string_buffer(bytes(msg), max(len(msg), 1))
        off = 0
        for s in (splits if splits is not None else [len(msg)]):
            self.lib.orc_beltCTRStepE(ctypes.byref(buf, off), _sz(s), st)
            off += s
        return buf.raw[: len(msg)]

    def ctr_start(self, key, iv):
        """(expanded key u32[8], ctr0 u32[4]) as bytes"""
        st = ctypes.create_string_buffer(32 + 16 + 16 + 8)
        self.lib.orc_beltCTRStart(st, bytes(key), _sz(len(key)), bytes(iv))
        return st.raw[:32], st.raw[32:48]

    def ctr_blocks_np(self, arr, key_words, ctr0_words, first=0, nthreads=1):
        assert arr.flags["C_CONTIGUOUS"] and arr.nbytes % 16 == 0
        self.lib.orc_beltCTR_blocks(ctypes.c_void_p(arr.ctypes.data), _sz(arr.nbytes // 16),
                                    bytes(key_words), bytes(ctr0_words), ctypes.c_uint64(first),
                                    ctypes.c_int(nthreads))

    def ecb(self, msg, key, decr=False):
        out = ctypes.create_string_buffer(max(len(msg), 1))
        code = self.lib.orc_beltECB(out, bytes(msg), _sz(len(msg)), bytes(key), _sz(len(key)), int(decr))
        return code, out.raw[: len(msg)]

    def cbc(self, msg, key, iv, decr=False):
        out = ctypes.create_string_buffer(max(len(msg), 1))
        code = self.lib.orc_beltCBC(out, bytes(msg), _sz(len(msg)), bytes(key), _sz(len(key)), bytes(iv), int(decr))
        return code, out.raw[: len(msg)]

    def bde(self, msg, key, iv, decr=False):
        out = ctypes.create_string_buffer(max(len(msg), 1))
        code = self.lib.orc_beltBDE(out, bytes(msg), _sz(len(msg)), bytes(key), _sz(len(key)), bytes(iv), int(decr))
        return code, out.raw[: len(msg)]

    # ---- belt-dwp (8f-2): ops = list of ("E"|"D"|"I"|"A", bytes) / ("G",) applied in order to one state;
    # returns (the E/D outputs concatenated in order, [mac after every "G"])
    def dwp_steps(self, key, iv, ops, mode="DWP"):
        f = lambda name: getattr(self.lib, f"orc_belt{mode}{name}")      # mode "DWP" or "CHE"
        st = ctypes.create_string_buffer(512)
        f("Start")(st, bytes(key), _sz(len(key)), bytes(iv))
        out, macs = b"", []
        for op in ops:
            if op[0] in "ED":
                b = ctypes.create_string_buffer(bytes(op[1]), max(len(op[1]), 1))
                f("StepE")(b, _sz(len(op[1])), st)
                out += b.raw[: len(op[1])]
            elif op[0] == "I":
                f("StepI")(bytes(op[1]), _sz(len(op[1])), st)
            elif op[0] == "A":
                f("StepA")(bytes(op[1]), _sz(len(op[1])), st)
            else:
                m = ctypes.create_string_buffer(8)
                f("StepG")(m, st)
                macs.append(m.raw)
        return out, macs

    def dwp_wrap(self, crit, open_, key, iv, mode="DWP"):
        dest, mac = ctypes.create_string_buffer(max(len(crit), 1)), ctypes.create_string_buffer(8)
        code = getattr(self.lib, f"orc_belt{mode}Wrap")(dest, mac, bytes(crit), _sz(len(crit)), bytes(open_), _sz(len(open_)),
                                        bytes(key), _sz(len(key)), bytes(iv))
        return code, dest.raw[: len(crit)], mac.raw

    def dwp_unwrap(self, crit, open_, mac, key, iv, mode="DWP"):
        dest = ctypes.create_string_buffer(max(len(crit), 1))
        code = getattr(self.lib, f"orc_belt{mode}Unwrap")(dest, bytes(crit), _sz(len(crit)), bytes(open_), _sz(len(open_)),
                                          bytes(mac), bytes(key), _sz(len(key)), bytes(iv))
        return code, dest.raw[: len(crit)]

    def bde_blocks_from(self, blocks, key, s_words, decr=False):
        """the BDE block loop from the tweak state s (bytes of 4 u32) -> (output, state after)"""
        buf = ctypes.create_string_buffer(bytes(blocks), max(len(blocks), 1))
        s = (ctypes.c_uint32 * 4).from_buffer_copy(bytes(s_words))
        self.lib.orc_beltBDE_blocks(buf, _sz(len(blocks) // 16), self.key_expand(key), s, int(decr))
        return buf.raw[: len(blocks)], bytes(s)

    def che_blocks_from(self, blocks, key, s_words):
        buf = ctypes.create_string_buffer(bytes(blocks), max(len(blocks), 1))
        s = (ctypes.c_uint32 * 4).from_buffer_copy(bytes(s_words))
        self.lib.orc_beltCHE_blocks(buf, _sz(len(blocks) // 16), self.key_expand(key), s)
        return buf.raw[: len(blocks)], bytes(s)

    def sde(self, msg, key, iv, decr=False):
        out = ctypes.create_string_buffer(max(len(msg), 1))
        code = self.lib.orc_beltSDE(out, bytes(msg), _sz(len(msg)), bytes(key), _sz(len(key)), bytes(iv), int(decr))
        return code, out.raw[: len(msg)]

    def wbl(self, msg, key, decr=False):
        buf = ctypes.create_string_buffer(bytes(msg), len(msg))
        code = self.lib.orc_beltWBL(buf, _sz(len(msg) // 16), self.key_expand(key), int(decr))
        return code, buf.raw[: len(msg)]

    def block_decr(self, block, key):
        w = (ctypes.c_uint32 * 4).from_buffer_copy(bytes(block))
        self.lib.orc_beltBlockDecr2(w, self.key_expand(key))
        return bytes(w)

    def mac(self, msg, key):
        out = ctypes.create_string_buffer(8)
        code = self.lib.orc_beltMAC(out, bytes(msg), _sz(len(msg)), bytes(key), _sz(len(key)))
        assert code == 0
        return out.raw

    def mac_steps(self, msg, key, splits):
        st = ctypes.create_string_buffer(32 + 16 + 16 + 16 + 8)
        self.lib.orc_beltMACStart(st, bytes(key), _sz(len(key)))
        off = 0
        for s in splits:
            self.lib.orc_beltMACStepA(bytes(msg[off:off + s]), _sz(s), st)
            off += s
        out = ctypes.create_string_buffer(8)
        self.lib.orc_beltMACStepG(out, st)
        return out.raw

    def belt_hash(self, msg):
        out = ctypes.create_string_buffer(32)
        self.lib.orc_beltHash(out, bytes(msg), _sz(len(msg)))
        return out.raw

    # --- bign
    def verify(self, h, s, p):
        return self.lib.orc_bign128Verify(bytes(h), bytes(s), bytes(p))

    def verify_rx(self, h, s, p):
        rx = ctypes.create_string_buffer(32)
        code = self.lib.orc_bign128Verify_ex(bytes(h), bytes(s), bytes(p), rx)
        return code, rx.raw

    def verify_l(self, l, oid, h, s, p):
        self.lib.orc_bignVerify_ex.restype = ctypes.c_uint32
        return self.lib.orc_bignVerify_ex(_sz(l), bytes(oid), _sz(len(oid)), bytes(h), bytes(s), bytes(p), None)

    def verify_batch_l(self, l, oid, hashes, sigs, pubs, nthreads=1):
        n = len(hashes) // (l // 4)
        codes = (ctypes.c_uint32 * max(n, 1))()
        self.lib.orc_bignVerify_batch(_sz(l), bytes(oid), _sz(len(oid)), bytes(hashes), bytes(sigs), bytes(pubs),
                                      _sz(n), codes, ctypes.c_int(nthreads))
        return list(codes)[:n]

    def pubkey_val(self, l, pub):
        self.lib.orc_bignPubkeyVal.restype = ctypes.c_uint32
        return self.lib.orc_bignPubkeyVal(_sz(l), bytes(pub))

    # ---- 8f-4 tail (oracle/bign_oracle.c): rng-driven entries take the rng's output stream
    def pubkey_calc(self, l, priv):
        out = ctypes.create_string_buffer(l // 2)
        self.lib.orc_bignPubkeyCalc.restype = ctypes.c_uint32
        return self.lib.orc_bignPubkeyCalc(_sz(l), out, bytes(priv)), out.raw

    def keypair_gen(self, l, rnd):
        no = l // 4
        priv = ctypes.create_string_buffer(no)
        pub = ctypes.create_string_buffer(2 * no)
        used = _sz(0)
        self.lib.orc_bignKeypairGen.restype = ctypes.c_uint32
        code = self.lib.orc_bignKeypairGen(_sz(l), priv, pub, bytes(rnd), _sz(len(rnd) // no), ctypes.byref(used))
        return code, priv.raw, pub.raw, used.value

    def sign2(self, l, oid, h, priv, t=None):
        sig = ctypes.create_string_buffer(3 * l // 8)
        self.lib.orc_bignSign2.restype = ctypes.c_uint32
        code = self.lib.orc_bignSign2(_sz(l), sig, bytes(oid), _sz(len(oid)), bytes(h), bytes(priv), t, _sz(len(t) if t else 0))
        return code, sig.raw

    def sign_rnd(self, l, oid, h, priv, rnd):
        no = l // 4
        sig = ctypes.create_string_buffer(3 * l // 8)
        used = _sz(0)
        self.lib.orc_bignSign_rnd.restype = ctypes.c_uint32
        code = self.lib.orc_bignSign_rnd(_sz(l), sig, bytes(oid), _sz(len(oid)), bytes(h), bytes(priv), bytes(rnd),
                                         _sz(len(rnd) // no), ctypes.byref(used))
        return code, sig.raw, used.value

    def pubkey_val_batch(self, l, pubs):
        n = len(pubs) // (l // 2)
        codes = (ctypes.c_uint32 * max(n, 1))()
        self.lib.orc_bignPubkeyVal_batch(_sz(l), bytes(pubs), _sz(n), codes)
        return list(codes)[:n]

    def verify_batch(self, hashes, sigs, pubs, nthreads=1):
        n = len(hashes) // 32
        codes = (ctypes.c_uint32 * n)()
        self.lib.orc_bign128Verify_batch(bytes(hashes), bytes(sigs), bytes(pubs), _sz(n), codes,
                                         ctypes.c_int(nthreads))
        return list(codes)

    # --- mixed
    def mixed_batch(self, msgs, msg_len, key, nthreads=1):
        n = len(msgs) // msg_len if msg_len else 0
        dig = ctypes.create_string_buffer(64 * n)
        tag = ctypes.create_string_buffer(8 * n)
        self.lib.orc_bash512_beltMAC_batch(bytes(msgs), _sz(msg_len), _sz(n), bytes(key),
                                           _sz(len(key)), dig, tag, ctypes.c_int(nthreads))
        return dig.raw, tag.raw

    def fill(self, nbytes, seed):
        buf = ctypes.create_string_buffer(nbytes)
        self.lib.orc_fill_splitmix64(buf, _sz(nbytes), ctypes.c_uint64(seed))
        return buf.raw

    def fill_np(self, arr, seed):
        self.lib.orc_fill_splitmix64(ctypes.c_void_p(arr.ctypes.data), _sz(arr.nbytes),
                                     ctypes.c_uint64(seed))


_cached = None


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(ORC_SO):
            build()
        _cached = Oracle(ctypes.CDLL(ORC_SO))
    return _cached


# the fixture reader lives in goldenlib (no oracle, no ctypes) so that bench.py's GPU legs can use the
# committed vectors without importing this module; re-exported here for the tests
from goldenlib import Golden  # noqa: E402,F401
