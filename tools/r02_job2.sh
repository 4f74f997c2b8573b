cd $GRAFT_REPO_ROOT
./tools/ubench/quad_dbl 32 > gpurun_out/quad_dbl.txt 2>&1
python tools/belt_ab.py > gpurun_out/belt_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for e in 11 13 15 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vsmall_$e -o v -- python $GRAFT_REPO_ROOT/tools/verify_small.py $e 20 > $GRAFT_REPO_ROOT/gpurun_out/vsmall_$e.log 2>&1
done
cd $GRAFT_REPO_ROOT
for e in 11 13 15 16; do echo "== 2^$e"; cat gpurun_out/vsmall_$e.log | tail -1; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/vsmall_$e/**/v_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("  %-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done > gpurun_out/vsmall_summary.txt 2>&1
bash tools/prof_belt.sh "0 4" 26 > gpurun_out/belt_pmc.txt 2>&1
tail -5 gpurun_out/quad_dbl.txt; tail -12 gpurun_out/belt_ab.txt; head -30 gpurun_out/vsmall_summary.txt
