#!/bin/bash
# one gpurun call: the one-signer / few-signers tests, first-call costs, the bench leg, the fuzz family
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ok; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bign_onekey.py tests/test_capi_exports.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python tools/keyed_first_call.py > $O/first_call.txt 2>&1; grep -v amdgpu.ids $O/first_call.txt
timeout 300 python bench.py --only verify --no-cpu --steps 10 --warmup 3 > $O/bench_verify.json 2> $O/bench_verify.err; echo "bench rc=$?"; tail -3 $O/bench_verify.err
python - <<P
import json
d=json.loads(open("$O/bench_verify.json").read().strip().splitlines()[-1])
for k in ("bignVerify_onekey","bignVerify_keyed"):
    v=d["others"][k]; print(k,{x:v.get(x) for x in ("value","ms_per_step","verdicts_as_expected","vs_general_entry")}, v["roofline"]["frac"])
P
FUZZ_FAMILIES=onekey timeout 200 python tests/fuzz_gpu.py 40 1401 2>&1 | tail -2
