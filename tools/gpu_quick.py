"""first-light GPU check: bashF + CTR vs the oracle, and a rough timing."""
import sys, os, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np
import torch
import bee2_amd, orclib

orc = orclib.load()
eng = bee2_amd.load()
print(eng.version(), torch.cuda.get_device_name(0))
# bashF
n = 1 << 20
host = np.empty(192 * n, dtype=np.uint8)
orc.fill_np(host, 0xBA5F)
dev = torch.from_numpy(host).cuda()
eng.bashF_batch_dev(dev); torch.cuda.synchronize()
got = dev.cpu().numpy()
want = host.copy(); orc.bashF_batch_np(want, nthreads=os.cpu_count())
print("bashF 2^20 parity:", bool((got == want).all()))
for _ in range(3): eng.bashF_batch_dev(dev)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): eng.bashF_batch_dev(dev)
torch.cuda.synchronize(); dt = (time.time() - t0) / 20
print(f"bashF: {dt*1e3:.3f} ms/launch  {n/dt/1e9:.2f} Gperm/s  {384*n/dt/1e12:.2f} TB/s")
# CTR
H = orc.beltH()
kw, c0 = orc.ctr_start(H[128:160], H[192:208])
nb = 1 << 26   # 1 GiB
buf = torch.randint(0, 256, (16 * nb,), dtype=torch.uint8, device="cuda")
src = buf[: 16 * 4096].cpu().numpy().copy(); src_tail = buf[-16 * 4096:].cpu().numpy().copy()
eng.beltCTR_blocks_dev(buf, kw, c0, 0); torch.cuda.synchronize()
w = src.copy(); orc.ctr_blocks_np(w, kw, c0, 0)
w2 = src_tail.copy(); orc.ctr_blocks_np(w2, kw, c0, nb - 4096)
print("CTR parity head/tail:", bool((buf[:16*4096].cpu().numpy() == w).all()), bool((buf[-16*4096:].cpu().numpy() == w2).all()))
for _ in range(2): eng.beltCTR_blocks_dev(buf, kw, c0, 0)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): eng.beltCTR_blocks_dev(buf, kw, c0, 0)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"CTR: {dt*1e3:.3f} ms/GiB  {1/dt:.1f} GiB/s")
# drop-in
print("bashF dropin A.2:", eng.bashF(H[:192]) == orc.bashF(H[:192]))
print("beltBlockEncr A.1:", eng.beltBlockEncr(H[:16], H[128:160]).hex())
ct, st = eng.beltCTR_steps(H[:48], H[128:160], H[192:208], [15, 7, 26])
print("CTR A.15:", ct.hex() == orc.ctr(H[:48], H[128:160], H[192:208]).hex(), ct.hex()[:32])
