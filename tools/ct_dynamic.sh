#!/bin/bash
# bash tools/ct_dynamic.sh [log2 n]  (on the GPU box): executed-instruction counters of the signing kernels per key class
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ct_dyn; rm -rf $O; mkdir -p $O
E=${1:-14}          # log2 of the batch: 18 = one lane per scalar with LDS look-ups (bign_mulbase_lds_kernel), 14 = one lane per scalar (scan), 10 = one wavefront per scalar (bign_mulbase_coop_kernel)
for c in random small ones sparse dense; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $O/$c -o b -- python $R/tools/ct_dynamic.py $c 128 $E > $O/$c.log 2>&1
done
cd $R
python - <<PY
import csv, collections, glob
tab = {}
for f in sorted(glob.glob("gpurun_out/ct_dyn/*/b_counter_collection.csv")):
    cls = f.split("/")[2]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bee2hip::", "")
        if "sign" not in k and "mulbase" not in k and "belt_hash_ragged" not in k: continue
        d[k][r["Counter_Name"]].append(int(float(r["Counter_Value"])))
    for k, v in d.items():
        tab.setdefault(k, {})[cls] = {c: x[-1] for c, x in v.items()}
for k, per in tab.items():
    print(k)
    names = sorted(next(iter(per.values())))
    for c in names:
        vals = {cls: per[cls].get(c) for cls in per}
        same = len(set(vals.values())) == 1
        print(f"   {c:<22s} {'IDENTICAL' if same else 'DIFFERS  '} " + " ".join(f"{cls}={v}" for cls, v in vals.items()))
PY
