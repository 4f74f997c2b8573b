#!/bin/bash
# bash tools/round_evidence.sh <tag>   (on the GPU box, ONE gpurun call): a round's closing evidence, everything under gpurun_out/evidence_<tag>/ --
#   gputests.log            python -m pytest tests -m gpu            (the parity tests proper, through the C ABI)
#   smoke.log               __graft_entry__.smoke()
#   bench.json / .txt       python bench.py --gpus 1 --steps 20 --warmup 5 as the driver calls it: the JSON line alone / the whole stdout; bench_detail.json
#   bench_all_detail.json   python bench.py --all (every leg's record)
#   bench_gpus2_gloo.json   python bench.py --gpus 2 self-launched over gloo on the one device (the N > 1 code path)
#   prof_summary/           tools/profile_headline.sh <tag>: rocprofv3 kernel stats + PMC passes per BASELINE launch
#   fuzz_*.txt              tests/fuzz_gpu.py on the product and the experiments build
# Copy what is to be judged into profiles/ afterwards (gpurun_out/ is scratch).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
cd $R
T0=$(date +%s); timeout 2400 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$? wall $(( $(date +%s) - T0 )) s" | tee -a $O/gputests.log; tail -4 $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
tail -1 $O/bench.txt > $O/bench.json; cp gpurun_out/bench_detail.json $O/bench_detail.json; grep "^\[bench\]" $O/bench.txt
T0=$(date +%s); timeout 900 python bench.py --all --steps 20 --warmup 5 > $O/bench_all.txt 2> $O/bench_all.err; echo "bench --all rc=$? wall $(( $(date +%s) - T0 )) s"
cp gpurun_out/bench_detail.json $O/bench_all_detail.json
BEE2_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --ctr-gib 4 > $O/bench_gpus2.txt 2> $O/bench_gpus2.err; echo "bench --gpus 2 (gloo, one device) rc=$?"
tail -1 $O/bench_gpus2.txt > $O/bench_gpus2_gloo.json
bash tools/profile_headline.sh $TAG > $O/prof.log 2>&1; tail -3 $O/prof.log
FUZZ_LIB=product timeout 700 python tests/fuzz_gpu.py ${FUZZ_S:-300} 605 > $O/fuzz_product.txt 2>&1; tail -3 $O/fuzz_product.txt
FUZZ_LIB=exp timeout 700 python tests/fuzz_gpu.py ${FUZZ_S:-300} 606 > $O/fuzz_exp.txt 2>&1; tail -3 $O/fuzz_exp.txt
du -sh $O
