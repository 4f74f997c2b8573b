"""t(one shard) vs t(four shards on four logical devices of one card) through bee2hip_bignVerifyL_batch_multi_dev: the margin behind
tests/test_gpu_multi.py::test_logical_devices_that_share_a_card_run_side_by_side.  usage (GPU box): python tools/ab/multi_ratio_probe.py [log2 shard]"""
import ctypes, os, sys, time
sys.path[:0]=['/root/repo','/root/repo/tests']
os.environ["BEE2HIP_FAKE_DEVICES"]="4"
import torch, goldenlib
from bee2_amd import engine as E
import bee2_amd
eng=bee2_amd.load(); eng.set_device(0)
G=goldenlib.Golden(); hs,ss,ps=G.bign_base_arrays(); m=1<<int(sys.argv[1]) if len(sys.argv)>1 else 1<<9
import numpy as np
dev=lambda b: torch.from_numpy(np.frombuffer(bytes(b),dtype=np.uint8).copy()).cuda()
th,tsg,tp=dev((hs*2)[:32*m]),dev((ss*2)[:48*m]),dev((ps*2)[:64*m])
vp=ctypes.c_void_p; _sz=ctypes.c_size_t; oid=E.LEVEL_OID[128]
def run(k):
    tc=[torch.full((m,),-1,dtype=torch.int32,device="cuda") for _ in range(k)]
    arr=lambda xs:(vp*k)(*[x.data_ptr() for x in xs]); CNT=(ctypes.c_size_t*k)(*([m]*k))
    args=(_sz(128),oid,_sz(11),arr([th]*k),arr([tsg]*k),arr([tp]*k),CNT,arr(tc),k)
    torch.cuda.synchronize(); eng.lib.bee2hip_bignVerifyL_batch_multi_dev(*args)
    best=1e9
    for _ in range(7):
        t0=time.perf_counter(); eng.lib.bee2hip_bignVerifyL_batch_multi_dev(*args); best=min(best,time.perf_counter()-t0)
    return best
for _ in range(4):
    t1,t4=run(1),run(4); print(round(t1*1e3,3),round(t4*1e3,3),round(t4/t1,2))
