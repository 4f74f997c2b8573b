#!/bin/bash
# bash tools/r05_final.sh <commit>  (on the GPU box, one gpurun call): the round's closing evidence on the final library -- the full -m gpu
# suite, smoke(), the driver-form bench runs (N = 1 with the CPU baseline; --gpus 2 self-launched over gloo on the one device), the
# constant-time counters of the signing side incl. the long-additional-input path, a fuzz campaign on both builds, rocprofv3 kernel stats +
# FETCH / WRITE + SQ counter passes of the bench command (condensed on the box).  Outputs: gpurun_out/r05z/, gpurun_out/prof_summary/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log; tail -4 $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
BEE2_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --ctr-gib 4 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2.err; echo "bench2 rc=$?"
timeout 900 bash tools/ct_dynamic.sh 14 > $O/ct_dynamic.txt 2>&1; grep -c IDENTICAL $O/ct_dynamic.txt; grep DIFFERS $O/ct_dynamic.txt | head
rm -rf $R/gpurun_out/ct_dyn
FUZZ_LIB=product timeout 400 python tests/fuzz_gpu.py 200 605 > $O/fuzz_product.txt 2>&1; tail -3 $O/fuzz_product.txt
FUZZ_LIB=exp timeout 400 python tests/fuzz_gpu.py 200 606 > $O/fuzz_exp.txt 2>&1; tail -3 $O/fuzz_exp.txt
PROF_COMMIT=$1 timeout 1500 bash tools/profile_round.sh r05 > $O/profile.log 2>&1; tail -2 $O/profile.log; ls $R/gpurun_out/prof_summary
du -sh $R/gpurun_out
