"""launch one beltCTR kernel variant a few times (for rocprofv3): python tools/ab/belt_run.py <variant> [log2 blocks] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402

v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 26
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = bee2_amd.load_experiments()
eng.set_device(0)
eng.lib.bee2hip_internal_tune(1, v)
kw, c0 = eng.beltCTRStart(bytes(range(32)), bytes(16))
st = torch.empty(16 << logn, dtype=torch.uint8, device="cuda")
st.random_(0, 256)
for _ in range(reps):
    eng.beltCTR_blocks_dev(st, kw, c0)
torch.cuda.synchronize()
