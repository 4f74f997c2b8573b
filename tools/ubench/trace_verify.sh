cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "$@"; do
if [ $L = lib ]; then P=bee2_amd/lib/libbee2hip.so; else P=tools/ubench/$L/libbee2hip.so; fi
rm -rf gpurun_out/prof_v_$L
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v_$L -o v -- env BEE2HIP_LIB=$P python bench.py --no-cpu --only verify --steps 3 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_v_$L/**/*kernel_stats.csv',recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if 'bign' in r['Name'] and 'gtable' not in r['Name'] and 'pubkey' not in r['Name']]
for r in sorted(rows,key=lambda r:r['Name']): print('$L', r['Name'].split('(')[0][-28:], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
PY
done
