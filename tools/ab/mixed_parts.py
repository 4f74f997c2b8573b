#!/usr/bin/env python3
"""configs[4] by its parts on the GPU box: the fused bash512 + beltMAC kernel, the same kernel hash-only and MAC-only, the plain
bashF batch kernel and the ECB block kernel over the same number of permutations / blocks.  usage: python tools/ab/mixed_parts.py [log2 n]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402

eng = bee2_amd.load(); eng.set_device(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ml = 4096
msgs = torch.empty(n * ml, dtype=torch.uint8, device="cuda")
msgs.view(torch.int64).random_()
key = bytes(range(32))
dig = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
tag = torch.empty(n * 8, dtype=torch.uint8, device="cuda")
kw, _ = eng.beltCTRStart(key, bytes(16))


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


both = t(lambda: eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, dig, tag, n))
d0, t0 = dig.clone(), tag.clone()
# the two halves as two kernels on two streams at once
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    ev = torch.cuda.Event(); ev.record()
    s1.wait_event(ev); s2.wait_event(ev)
    with torch.cuda.stream(s1):
        eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, dig, None, n)
    with torch.cuda.stream(s2):
        eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, None, tag, n)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)


dig.zero_(); tag.zero_()
twok = t(two)
assert torch.equal(d0, dig) and torch.equal(t0, tag)
print(f"hash-only and MAC-only kernels on two streams at once {twok:7.3f} ms   {n / twok / 1e3:8.1f} M msg/s")
honly = t(lambda: eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, dig, None, n))
monly = t(lambda: eng.bashHash_beltMAC_batch_dev(msgs, ml, 256, key, None, tag, n))
states = msgs[: (n * ml) // 192 * 192]
perms = states.numel() // 192
bf = t(lambda: eng.bashF_batch_dev(states))
ecb = t(lambda: eng.beltModes_blocks_dev(0, msgs, msgs, kw))
print(f"n = {n} messages of 4 KiB")
print(f"fused hash + MAC          {both:8.3f} ms   {n / both / 1e3:8.1f} M msg/s")
print(f"fused kernel, hash only   {honly:8.3f} ms   {n / honly / 1e3:8.1f} M msg/s   ({65 * n / honly / 1e6:6.2f} G perm/s)")
print(f"fused kernel, MAC only    {monly:8.3f} ms   {n / monly / 1e3:8.1f} M msg/s   ({257 * n / monly / 1e6:6.2f} G blocks/s)")
print(f"hash only + MAC only      {honly + monly:8.3f} ms   {n / (honly + monly) / 1e3:8.1f} M msg/s")
print(f"bashF batch kernel, {perms} states {bf:8.3f} ms  ({perms / bf / 1e6:6.2f} G perm/s)  -> 65 perms/msg: {perms / bf / 65 / 1e3:8.1f} M msg/s")
print(f"ECB block kernel, {n * ml // 16} blocks {ecb:8.3f} ms  ({n * ml / 16 / ecb / 1e6:6.2f} G blocks/s) -> 257 blocks/msg: {n * ml / 16 / ecb / 257 / 1e3:8.1f} M msg/s")
