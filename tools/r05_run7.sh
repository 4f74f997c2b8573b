#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests/test_gpu_bign_onekey.py tests/test_gpu_threads.py tests/test_gpu_multi.py tests/test_gpu_oom_injection.py tests/test_gpu_bign_sign.py -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
timeout 900 bash tools/ct_dynamic.sh 14 > $O/ct_dynamic.txt 2>&1; grep -c IDENTICAL $O/ct_dynamic.txt; grep DIFFERS $O/ct_dynamic.txt | head; grep -A9 belt_hash_ragged $O/ct_dynamic.txt | head -12
rm -rf gpurun_out/ct_dyn
python bench.py --steps 20 --warmup 5 --only bashF,verify > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g/bench.json').read().strip().splitlines()[-1])
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if 'bashF' in k or k in ('frac','avg_launch_ms')})
PY
