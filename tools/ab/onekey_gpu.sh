#!/bin/bash
# one gpurun call: the one-signer / few-signers tests, sizes (one lane / four lanes per signature forced), the bench leg, the fuzz family
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ok; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bign_onekey.py tests/test_capi_exports.py -x -q > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 600 python tools/ab/onekey_sizes.py 20 > $O/sizes.txt 2>&1; grep -v amdgpu.ids $O/sizes.txt
FUZZ_FAMILIES=onekey FUZZ_LIB=exp timeout 200 python tests/fuzz_gpu.py 40 1501 2>&1 | tail -2
FUZZ_FAMILIES=onekey timeout 200 python tests/fuzz_gpu.py 30 1502 2>&1 | tail -2
