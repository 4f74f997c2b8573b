#!/bin/bash
# round 5, first GPU call: the new tests, the bench line with the strong split, per-kernel times of the verification floor
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests/test_gpu_mixed.py -m gpu -x -q -k "config4_whole" > $O/t_mixed.log 2>&1; tail -3 $O/t_mixed.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python -m pytest tests/test_bench_launch.py -m gpu -x -q > $O/t_bench.log 2>&1; tail -3 $O/t_bench.log
for spec in "15 0" "15 0x43" "16 0" "16 0x23" "14 0" "17 0"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1_$2 -o v -- python tools/verify_floor_probe.py $1 $2 40 > $O/probe_$1_$2.log 2>&1
  tail -1 $O/probe_$1_$2.log
  python - <<PY
import csv,glob
f=glob.glob('$O/prof_$1_$2/**/*kernel_stats.csv',recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if 'bign' in r['Name'] and 'gtable' not in r['Name']]
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])): print('   ', r['Name'].split('(')[0][-44:], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
PY
done 2>&1 | tee $O/verify_floor.txt
python tools/size_sweep.py > $O/size_sweep.txt 2>&1; tail -30 $O/size_sweep.txt
