#!/bin/bash
# one gpurun call: the one-signer tests, its sizes table with the 8-bit / 16-bit table of the key, the bench leg
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ok; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bign_onekey.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
ONEKEY_TAB16=63 timeout 600 python tools/onekey_sizes.py 20 > $O/sizes_tab8.txt 2>&1; grep -v amdgpu.ids $O/sizes_tab8.txt
ONEKEY_TAB16=0 timeout 600 python tools/onekey_sizes.py 20 > $O/sizes_tab16.txt 2>&1; grep -v amdgpu.ids $O/sizes_tab16.txt
timeout 300 python bench.py --only verify --no-cpu --steps 10 --warmup 3 > $O/bench_verify.json 2> $O/bench_verify.err; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open("$O/bench_verify.json").read().strip().splitlines()[-1])
v=d["others"]["bignVerify_onekey"]; print({x:v.get(x) for x in ("value","ms_per_step","verdicts_as_expected","vs_general_entry")}, v["roofline"]["frac"])
P
