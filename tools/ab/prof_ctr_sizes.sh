#!/bin/bash
# Why is a 16 GiB beltCTR pass ~8 % slower per byte than a 1 GiB pass?  Memory-side counters of the product kernel at both
# sizes (separate --pmc passes, kernel-trace only):  bash tools/ab/prof_ctr_sizes.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03/ctr_sizes
rm -rf $O; mkdir -p $O
for logn in 26 30; do
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
             "TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" \
             "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY"; do
    rocprofv3 --pmc $set --output-format csv -d $O/n${logn}_s$i -o b -- python $R/tools/ab/belt_run.py 0 $logn 6 > $O/n${logn}_s$i.log 2>&1
    i=$((i+1))
  done
done
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/r03/ctr_sizes/n*/b_counter_collection.csv")):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        if "beltCTR" not in k: continue
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"])); d[k]["dur_ns"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in d.items():
        print(f.split("/")[3], {c: round(sum(x[2:])/len(x[2:]),1) for c,x in v.items()})
for f in sorted(glob.glob("gpurun_out/r03/ctr_sizes/*.log")):
    t=open(f).read()
    if "rror" in t: print(f, t[-400:])
PY
