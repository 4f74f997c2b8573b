#!/bin/bash
# round 5, second GPU call: the bench line, its tests, SQ counters of the verification kernels at 1 / 2 wavefronts per SIMD
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python -m pytest tests/test_bench_launch.py -m gpu -x -q > $O/t_bench.log 2>&1; tail -3 $O/t_bench.log
for spec in "14 0" "15 0" "15 0x43" "16 0" "16 0x23" "16 0x43" "18 0"; do
  set -- $spec
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_$1_$2 -o v -- python tools/verify_floor_probe.py $1 $2 10 > $O/pmcprobe_$1_$2.log 2>&1
  echo "== n = 2^$1 form $2"
  python - <<PY
import csv,glob,collections
f=glob.glob('$O/pmc_$1_$2/**/*counter_collection.csv',recursive=True)
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('bee2hip::','')
    if 'bign' not in k or 'gtable' in k: continue
    d[k][r['Counter_Name']].append(float(r['Counter_Value'])); d[k]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3/ (1 if r['Counter_Name']=='SQ_WAVES' else 1))
for k,v in d.items():
    a={c: sum(x)/len(x) for c,x in v.items()}
    busy=a.get('SQ_BUSY_CYCLES',0)
    print('   %-44s dur %.1f us  waves %.0f  valu_busy %.3f  insts_valu/wave %.0f  wave_cycles/busy %.2f' % (k[-44:], a['dur_us'], a.get('SQ_WAVES',0), a.get('SQ_ACTIVE_INST_VALU',0)/(8*busy) if busy else 0, a.get('SQ_INSTS_VALU',0)/max(1,a.get('SQ_WAVES',1)), a.get('SQ_WAVE_CYCLES',0)*4/(busy*32) if busy else 0))
PY
done 2>&1 | tee $O/verify_pmc.txt
