#!/usr/bin/env python3
"""One-off pinning run (build container): oracle vs the reference on >= 1e5 random items
per primitive.  Prints one line per primitive; result recorded in DESIGN.md."""
import ctypes
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tools")]
import orclib  # noqa: E402
import refgen  # noqa: E402

_sz = ctypes.c_size_t


def main(n_verify=100_000):
    orc, L = orclib.load(), refgen.ref()
    t0 = time.time()
    n = 200_000
    data = orc.fill(192 * n, 1)
    ref = ctypes.create_string_buffer(data, len(data))
    base = ctypes.addressof(ref)
    for i in range(n):
        L.bashF(ctypes.c_void_p(base + 192 * i), None)
    print(f"bashF   {n} states   equal={orc.bashF_batch(data, 8) == ref.raw}  {time.time()-t0:.1f}s")
    t0 = time.time()
    H = orc.beltH()
    data = orc.fill(16 * 1_000_000 + 7, 2)
    out = ctypes.create_string_buffer(len(data))
    L.beltCTR(out, data, _sz(len(data)), H[128:160], _sz(32), H[192:208])
    print(f"beltCTR 1e6 blocks     equal={orc.ctr(data, H[128:160], H[192:208]) == out.raw}  {time.time()-t0:.1f}s")
    t0 = time.time()
    rnd = random.Random(5)
    ok, nblk = True, 0
    for _ in range(400):                                   # belt-bde (8f-1): lengths 1..2000 blocks, all key sizes
        nb = rnd.choice((1, 2, 3, 63, 64, 65, 127, 128, 129)) if rnd.random() < 0.5 else rnd.randrange(1, 2000)
        msg, key, iv = rnd.randbytes(16 * nb), rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        for fn, decr in (("beltBDEEncr", False), ("beltBDEDecr", True)):
            out = ctypes.create_string_buffer(len(msg))
            assert getattr(L, fn)(out, msg, _sz(len(msg)), key, _sz(len(key)), iv) == 0
            ok = ok and orc.bde(msg, key, iv, decr) == (0, out.raw)
        nblk += 2 * nb
    print(f"beltBDE {nblk} blocks     equal={ok}  {time.time()-t0:.1f}s")
    t0 = time.time()
    rnd = random.Random(19)
    ok = True
    for _ in range(3000):                                  # belt-dwp (8f-2): wrap, unwrap (incl. bad mac)
        crit = rnd.randbytes(rnd.choice((0, 1, 7, 15, 16, 17, 31, 32, 33, 100, rnd.randrange(0, 600))))
        op = rnd.randbytes(rnd.choice((0, 1, 15, 16, 17, 32, 47, rnd.randrange(0, 300))))
        key, iv = rnd.randbytes(rnd.choice((16, 24, 32))), rnd.randbytes(16)
        d, m = ctypes.create_string_buffer(max(len(crit), 1)), ctypes.create_string_buffer(8)
        assert L.beltDWPWrap(d, m, crit, _sz(len(crit)), op, _sz(len(op)), key, _sz(len(key)), iv) == 0
        ct = d.raw[: len(crit)]
        ok = ok and orc.dwp_wrap(crit, op, key, iv) == (0, ct, m.raw)
        mac = m.raw if rnd.random() < 0.7 else bytes([m.raw[0] ^ 1]) + m.raw[1:]
        d2 = ctypes.create_string_buffer(max(len(crit), 1))
        rc = L.beltDWPUnwrap(d2, ct, _sz(len(ct)), op, _sz(len(op)), mac, key, _sz(len(key)), iv)
        oc, od = orc.dwp_unwrap(ct, op, mac, key, iv)
        ok = ok and rc == oc and (rc != 0 or od == d2.raw[: len(crit)] == crit)
    print(f"beltDWP 3000 wrap/unwrap  equal={ok}  {time.time()-t0:.1f}s   (step sequences with mid-stream "
          f"StepG: tests/test_oracle_vs_ref.py)")
    t0 = time.time()
    rnd = random.Random(3)
    base_tr = refgen.make_triples(4096, 0xB164)
    hs, ss, ps, want = bytearray(), bytearray(), bytearray(), []
    for i in range(n_verify):
        h, s, p = (bytearray(x) for x in base_tr[i % len(base_tr)])
        kind = rnd.randrange(8)
        if kind == 1:
            s[rnd.randrange(16)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            s[16 + rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        elif kind == 3:
            h[rnd.randrange(32)] ^= 1 << rnd.randrange(8)
        elif kind == 4:
            p[rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        hs += h; ss += s; ps += p
        want.append(refgen.verify(bytes(h), bytes(s), bytes(p)))
    got = orc.verify_batch(hs, ss, ps, nthreads=8)
    from collections import Counter
    print(f"verify  {n_verify} sigs    equal={got == want}  codes={dict(Counter(want))}  {time.time()-t0:.1f}s")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100_000)
