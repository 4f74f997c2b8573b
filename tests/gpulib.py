"""helpers shared by the -m gpu tests (product = bee2_amd via the C ABI; checker = oracle)"""
import numpy as np
import torch

import bee2_amd


def engine():
    eng = bee2_amd.load()          # raises if libbee2hip.so is missing: no fallback
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return eng


def exp_engine():
    """libbee2hip_exp.so (-DBEE2HIP_EXPERIMENTS): the same kernels plus the hooks of include/bee2hip_internal.h -- field
    arithmetic self-test, forced kernel choices, fault injection.  Only tests that need a hook use it."""
    eng = bee2_amd.load_experiments()
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    eng.set_device(torch.cuda.current_device())
    return eng


def dev(b):
    """bytes / numpy uint8 -> uint8 CUDA tensor (copy)"""
    if isinstance(b, (bytes, bytearray)):
        b = np.frombuffer(bytes(b), dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(b).copy()).cuda()


def host(t):
    return t.cpu().numpy().tobytes()
