#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -5 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c/bench.json').read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('metric','value','ms_per_step')}))
print(json.dumps(d['roofline'])[:3000])
print(json.dumps(d['cpu_baseline'])[:3000])
PY
