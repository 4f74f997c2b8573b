#!/bin/bash
# A/B of the ask-ahead loads (prep / inv / main29 / quad-pair comb) against round 4's kernels, alternating in one call; then the bign tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
for i in 1 2 3; do
  BEE2HIP_LIB=tools/ubench/base/libbee2hip.so python tools/verify_sizes.py base 2>/dev/null
  python tools/verify_sizes.py new 2>/dev/null
done | tee $O/verify_sizes_ab.txt
python -m pytest tests/test_gpu_bign.py tests/test_gpu_bign_onekey.py tests/test_gpu_hostpath.py tests/test_gpu_reftests.py -m gpu -x -q > $O/gputests.log 2>&1; tail -4 $O/gputests.log
