#!/bin/bash
# usage: tools/ab/pmc_one.sh <tag> "<bench args>" "<counters set 1>" ["<counters set 2>" ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ARGS=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
i=0
for set in "$@"; do
  rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -o b -- python bench.py --no-cpu $ARGS > $OUT/s$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/s*/b_counter_collection.csv")):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-36:]
        if "rocclr" in k or "at::" in k or "elementwise" in k: continue
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"])); d[k]["dur_ns"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in d.items():
        print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
