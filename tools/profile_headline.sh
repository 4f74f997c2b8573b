#!/bin/bash
# bash tools/profile_headline.sh [tag]   (on the GPU box, inside one gpurun call)
# rocprofv3 evidence at the HEADLINE launch sizes and only there: for each BASELINE leg (bashF 2^20 states, beltCTR 16 GiB,
# bignVerify 2^18, bash512+beltMAC 2^21 x 4 KiB) one `bench.py --only <leg> --headline-only` run per pass --
#   stats : --kernel-trace --stats           (per-kernel average duration; must agree with bench.py's hipEvent figure)
#   fetch : --pmc FETCH_SIZE                 } separate passes, as MI355X_MICROARCH.md prescribes for HBM traffic
#   write : --pmc WRITE_SIZE                 }
#   sq1   : VALU side (valu_busy = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024))
#   sq2   : LDS side (lds_array_busy = SQ_LDS_IDX_ACTIVE / (256 x SQ_BUSY_CYCLES / 32), bank conflicts)
# Counters in their own runs (no trace domains with --pmc).  Condensed on the box by tools/summarize_headline.py into
# gpurun_out/prof_summary/<tag>_pmc_headline.json + <tag>_kernel_stats_<leg>.csv; copy those into profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$R/gpurun_out/prof_headline
rm -rf $OUT; mkdir -p $OUT
cd $R
LEGS=${LEGS:-"bashF ctr verify mixed"}
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
for leg in $LEGS; do
  CMD="python bench.py --only $leg --headline-only --no-cpu --steps 10 --warmup 2"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${leg}_stats -o b -- $CMD > $OUT/${leg}_stats.log 2>&1; echo "$leg stats rc=$?"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${leg}_fetch -o b -- $CMD > $OUT/${leg}_fetch.log 2>&1; echo "$leg fetch rc=$?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${leg}_write -o b -- $CMD > $OUT/${leg}_write.log 2>&1; echo "$leg write rc=$?"
  timeout 600 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/${leg}_sq1 -o b -- $CMD > $OUT/${leg}_sq1.log 2>&1; echo "$leg sq1 rc=$?"
  timeout 600 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/${leg}_sq2 -o b -- $CMD > $OUT/${leg}_sq2.log 2>&1; echo "$leg sq2 rc=$?"
done
PROF_DST=$R/gpurun_out/prof_summary python tools/summarize_headline.py $TAG > $OUT/summarize.log 2>&1; tail -40 $OUT/summarize.log
cp $OUT/*.log $R/gpurun_out/prof_summary/ 2>/dev/null
du -sh $OUT; rm -rf $OUT
