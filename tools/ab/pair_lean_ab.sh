cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_bign.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python tools/ab/verify_mid_ab.py 16 2>&1 | grep -v amdgpu | head -9
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc -o b -- python tools/ab/verify_floor_probe.py 15 0 20 > $O/pmc.log 2>&1
python - <<PY
import csv, collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/pmc/b_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0].replace("void bee2hip::","")[:44]
    if "quad29" not in k: continue
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"])); d[k]["dur"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    print(f"   {k:44s} dur {m['dur']/1e3:7.1f} us waves {m['SQ_WAVES']:.0f} valu_busy {m['SQ_ACTIVE_INST_VALU']*4/(m['SQ_BUSY_CYCLES']/32*1024):.3f} insts_valu/wave {m['SQ_INSTS_VALU']/max(1,m['SQ_WAVES']):.0f}")
PY
rm -rf $O/pmc
