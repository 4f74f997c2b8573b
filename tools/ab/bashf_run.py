"""launch one bashF kernel variant a few times (for rocprofv3): python tools/ab/bashf_run.py <variant> [log2 n] [reps]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bee2_amd  # noqa: E402

v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
eng = bee2_amd.load_experiments()
eng.set_device(0)
eng.lib.bee2hip_internal_tune(0, v)
n = 1 << logn
st = torch.empty(192 * n, dtype=torch.uint8, device="cuda")
st.random_(0, 256)
for _ in range(reps):
    eng.bashF_batch_dev(st)
torch.cuda.synchronize()
