#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests/test_gpu_oom_injection.py tests/test_gpu_multi.py tests/test_bench_launch.py tests/test_gpu_bign_sign.py tests/test_gpu_mixed.py tests/test_gpu_bign_onekey.py tests/test_gpu_graphs.py -m gpu -x -q > $O/gputests.log 2>&1; tail -5 $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
bash tools/profile_round.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
