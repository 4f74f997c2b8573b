import ctypes, json, os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import bee2_amd
from bee2_amd.engine import bign_params
eng = bee2_amd.load(); eng.set_device(0)
FIX = json.load(open("/root/repo/tests/golden/bign_generic.json"))
for ci, c in enumerate(FIX["curves"]):
    prm = bign_params(); prm.l = c["l"]
    for f in ("p", "a", "b", "q", "yG"):
        raw = bytes.fromhex(c[f]); ctypes.memmove(getattr(prm, f), raw + bytes(64 - len(raw)), 64)
    x = [y for y in FIX["cases"] if y["curve"] == ci and y["name"] == "good"][0]
    h, s, k = (bytes.fromhex(x[f]) for f in ("hash", "sig", "pubkey"))
    for n in (1, 4096):
        eng.bignVerify_batch(h * n, s * n, k * n, oid_der=bytes.fromhex(x["oid"]), params=prm)
        t = time.perf_counter(); code, got = eng.bignVerify_batch(h * n, s * n, k * n, oid_der=bytes.fromhex(x["oid"]), params=prm); dt = time.perf_counter() - t
        assert code == 0 and set(got) == {0}
        print(f"{c['kind']} l={c['l']} n={n}: {dt*1e3:.1f} ms, {n/dt:.0f} verifies/s")
