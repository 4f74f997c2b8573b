#!/bin/bash
# PMC counters of beltCTR kernel variants: bash tools/ab/prof_belt.sh "<variants>" <log2 blocks>   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/belt_pmc
rm -rf $O; mkdir -p $O
VARS=${1:-"0 4"}; LOGN=${2:-26}
for v in $VARS; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE"; do
    rocprofv3 --pmc $set --output-format csv -d $O/v${v}_s$i -o b -- python $R/tools/ab/belt_run.py $v $LOGN 6 > $O/v${v}_s$i.log 2>&1
    i=$((i+1))
  done
done
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/belt_pmc/v*/b_counter_collection.csv")):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-60:]
        if "beltCTR" not in k: continue
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"])); d[k]["dur_ns"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in d.items():
        print(f.split("/")[2], k, {c: round(sum(x[2:])/len(x[2:]),1) for c,x in v.items()})
PY
