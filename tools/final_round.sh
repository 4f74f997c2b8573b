#!/bin/bash
# bash tools/final_round.sh  (on the GPU box, one gpurun call): the round's closing evidence -- the full -m gpu suite, smoke(), the driver-form bench
# runs (N = 1 with the CPU baseline; --gpus 2 self-launched over gloo on the one device), rocprofv3 kernel stats + HBM-traffic counters of the
# same bench command (tools/profile_round.sh, every rocprofv3 under a timeout), and a fuzz campaign on both builds.  Outputs: gpurun_out/fin4/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fin4; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log; tail -4 $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
BEE2_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --ctr-gib 4 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2.err; echo "bench2 rc=$?"
# rocprofv3: kernel stats of the bench command, bashF alone, FETCH / WRITE passes (each under a timeout; csv output)
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof; rm -rf $P; mkdir -p $P
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- $CMD > $P/stats.log 2>&1; echo "stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_bashF -o bench -- $CMD --only bashF --headline-only > $P/stats_bashF.log 2>&1; echo "stats_bashF rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $P/pmc_$c -o bench -- $CMD --only bashF,ctr,verify --ctr-gib 4 --headline-only > $P/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $R
FUZZ_LIB=product timeout 400 python tests/fuzz_gpu.py 150 505 > $O/fuzz_product.txt 2>&1; tail -3 $O/fuzz_product.txt
FUZZ_LIB=exp timeout 400 python tests/fuzz_gpu.py 150 506 > $O/fuzz_exp.txt 2>&1; tail -3 $O/fuzz_exp.txt
find $P -name '*.csv' | head -20; du -sh $O
