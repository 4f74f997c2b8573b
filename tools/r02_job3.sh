cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 13 14 15; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vsmallq_$e -o v -- python $R/tools/verify_small.py $e 20 > $R/gpurun_out/vsmallq_$e.log 2>&1
done
cd $R
for e in 13 14 15; do echo "== 2^$e"; tail -1 gpurun_out/vsmallq_$e.log; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/vsmallq_$e/**/v_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("  %-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
