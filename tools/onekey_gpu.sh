cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ok
timeout 900 python -m pytest tests/test_gpu_bign_onekey.py -x -q > gpurun_out/ok/tests.log 2>&1; tail -15 gpurun_out/ok/tests.log
timeout 300 python bench.py --only verify --no-cpu --steps 10 --warmup 3 > gpurun_out/ok/bench_verify.json 2> gpurun_out/ok/bench_verify.err; echo "bench rc=$?"; tail -3 gpurun_out/ok/bench_verify.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/ok/bench_verify.json').read().strip().splitlines()[-1])
o=d.get('others',{}); 
for k in ('bignVerify','bignVerify_onekey'):
    v=o.get(k) or (d if d.get('metric','').startswith('bign') else {})
    print(k, {x:v.get(x) for x in ('value','ms_per_step','verdicts_as_expected','vs_general_entry')}, v.get('roofline',{}).get('frac'))
P
timeout 300 python tools/verify_mid_ab.py > gpurun_out/ok/verify_mid_ab.txt 2>&1; cat gpurun_out/ok/verify_mid_ab.txt
