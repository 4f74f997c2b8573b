"""GPU check of the GF(p) layer and batched verify against Python ints / golden / oracle."""
import sys, os, time, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, ctypes
import bee2_amd, orclib
orc = orclib.load(); eng = bee2_amd.load(); G = orclib.Golden()
P = 2**256 - 189
rnd = random.Random(1)
vals = [0, 1, 2, P-1, P, P+1, 2**256-1, 2**256-189, 2**255, 189, 188] + [rnd.getrandbits(256) for _ in range(500)]
A = [rnd.choice(vals) for _ in range(4096)]; B = [rnd.choice(vals) for _ in range(4096)]
def to_t(xs): return torch.from_numpy(np.frombuffer(b"".join(x.to_bytes(32,"little") for x in xs), dtype=np.uint8).copy()).cuda()
def from_t(t): raw = t.cpu().numpy().tobytes(); return [int.from_bytes(raw[i:i+32],"little") for i in range(0,len(raw),32)]
ta, tb = to_t(A), to_t(B); out = torch.empty_like(ta)
ops = {0: lambda a,b: a*b%P, 1: lambda a,b: a*a%P, 2: lambda a,b:(a+b)%P, 3: lambda a,b:(a-b)%P, 4: lambda a,b: pow(a,P-2,P), 5: lambda a,b: 3*a*b%P, 6: lambda a,b: 8*a*a%P, 7: lambda a,b: a%P}
for op, f in ops.items():
    code = eng.lib.bee2hip_debug_fe(op, ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_size_t(len(A)), None)
    torch.cuda.synchronize()
    got = from_t(out); want = [f(a,b) for a,b in zip(A,B)]
    bad = [i for i in range(len(A)) if got[i]!=want[i]]
    print("fe op", op, "code", code, "mismatches", len(bad), (hex(A[bad[0]]), hex(B[bad[0]]), hex(got[bad[0]]), hex(want[bad[0]])) if bad else "")
# verify: golden base
hs, ss, ps = G.bign_base_arrays()
def dev(b): return torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).cuda()
n = len(hs)//32
codes = torch.zeros(n, dtype=torch.int32, device="cuda")
t0=time.time(); eng.bign128Verify_batch_dev(dev(hs), dev(ss), dev(ps), codes); torch.cuda.synchronize(); print("first call (table build) %.3fs"%(time.time()-t0))
c = codes.cpu().numpy(); print("base: n", n, "nonzero codes", int((c!=0).sum()), c[:8])
# edge cases
E = G.bign_edge
eh = b"".join(bytes.fromhex(e["hash"]) for e in E); es = b"".join(bytes.fromhex(e["sig"]) for e in E); ep = b"".join(bytes.fromhex(e["pubkey"]) for e in E)
codes = torch.zeros(len(E), dtype=torch.int32, device="cuda")
eng.bign128Verify_batch_dev(dev(eh), dev(es), dev(ep), codes); torch.cuda.synchronize()
c = codes.cpu().numpy().astype(np.int64) & 0xffffffff
bad = [(E[i]["name"], int(c[i]), E[i]["code"]) for i in range(len(E)) if int(c[i]) != E[i]["code"]]
print("edge: n", len(E), "mismatches", len(bad), bad[:10])
# timing at 2^18 by tiling
reps = (1<<18)//n
H, S, PK = dev(hs*reps), dev(ss*reps), dev(ps*reps)
N = n*reps; codes = torch.zeros(N, dtype=torch.int32, device="cuda")
eng.bign128Verify_batch_dev(H,S,PK,codes); torch.cuda.synchronize()
t0=time.time(); eng.bign128Verify_batch_dev(H,S,PK,codes); torch.cuda.synchronize(); dt=time.time()-t0
print(f"verify 2^18: {dt*1e3:.2f} ms  {N/dt/1e6:.2f} M verifies/s; all ok: {bool((codes==0).all())}")
