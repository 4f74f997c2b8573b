#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
for i in 1 2; do
  BEE2HIP_LIB=tools/ubench/base/libbee2hip.so python tools/verify_sizes.py base 2>/dev/null
  python tools/verify_sizes.py new 2>/dev/null
done | tee $O/verify_sizes_ab.txt
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -4 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
PROF_COMMIT=$1 bash tools/profile_round.sh r05 > $O/profile.log 2>&1; tail -3 $O/profile.log; ls gpurun_out/prof_summary
