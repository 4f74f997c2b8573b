cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
python tools/ab/verify_mid_ab.py 19 > $O/forms.txt 2>&1; cat $O/forms.txt | tail -20
for spec in "14 0" "15 0" "16 0" "18 2" "18 1"; do set -- $spec
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc_$1_$2 -o b -- python tools/ab/verify_floor_probe.py $1 $2 20 > $O/pmc_$1_$2.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$1_$2 -o b -- python tools/ab/verify_floor_probe.py $1 $2 40 > $O/st_$1_$2.log 2>&1
  python - <<PY
import csv, collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/pmc_$1_$2/b_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0].replace("void bee2hip::","")[:44]
    if "at::" in k or "rocclr" in k or "elementwise" in k: continue
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"])); d[k]["dur"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("== n = 2^$1 form $2")
for k,v in d.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    if not m.get("SQ_BUSY_CYCLES"): continue
    print(f"   {k:44s} dur {m['dur']/1e3:7.1f} us waves {m['SQ_WAVES']:.0f} valu_busy {m['SQ_ACTIVE_INST_VALU']*4/(m['SQ_BUSY_CYCLES']/32*1024):.3f} insts_valu/wave {m['SQ_INSTS_VALU']/max(1,m['SQ_WAVES']):.0f}")
for r in csv.DictReader(open("$O/st_$1_$2/b_kernel_stats.csv")):
    print("   stats", r["Name"].split("(")[0][-40:], r["Calls"], f"{float(r['AverageNs'])/1e3:.1f} us")
PY
done > $O/counters.txt 2>&1
cat $O/counters.txt
rm -rf $O/pmc_* $O/st_*
