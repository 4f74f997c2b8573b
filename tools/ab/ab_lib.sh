#!/bin/bash
# A/B two builds of libbee2hip.so on the GPU box, alternating runs inside ONE gpurun call
# (boxes differ by +-3 %, so numbers from different calls are not comparable).
#   build the variant here, e.g.:
#       make -C bee2_amd/csrc EXTRA=-DSOMETHING OUT=$PWD/tools/ubench/variant
#   or keep the old build as the baseline before rebuilding in-tree:
#       mkdir -p tools/ubench/base && cp bee2_amd/lib/libbee2hip.so tools/ubench/base/
#   run on the GPU:  bash tools/ab/ab_lib.sh <libA> <libB> <workload: bashF|ctr|verify|mixed|modes> [pytest -k filter]
# Prints wall-clock value and, where bench.py reports one, the event-timed kernel rate for A and B,
# three alternations, then runs the parity tests selected by the filter against B
# (a variant that is fast and wrong is worth nothing).
A=${1:?lib A}; B=${2:?lib B}; W=${3:?workload}; K=${4:-$W}
run() { BEE2HIP_LIB=$2 python bench.py --no-cpu --only $W ${STEPS:+--steps $STEPS} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
line='$1 %-9s %.4g %s' % (d['metric'][:24], d['value'], d['unit'])
r=d.get('roofline') or {}
if 'avg_launch_ms' in r: line+='  kernel %.4f ms' % r['avg_launch_ms']
std={'value','n_gpus','steps','warmup','ms_per_step','vs_baseline'}
for k,v in d.items():
    if isinstance(v,float) and k not in std: line+='  %s %.4g' % (k, v)
for k,v in (d.get('others') or {}).items(): line+='  | %s %.4g' % (k, v['value'])
print(line)"; }
for i in 1 2 3; do run A $A; run B $B; done
echo "parity, B:"; BEE2HIP_LIB=$B python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -1
