#!/bin/bash
# rocprofv3 evidence for a round (run on the GPU box via gpurun; outputs under gpurun_out/prof;
# then here: python tools/summarize_prof.py rNN)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
# the headline workload alone: the full run also launches the bashF kernel on single states (drop-in latency leg, sponge
# finalisation), which would pull the per-kernel average away from the batch launches the roofline is about
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bashF -o bench -- $CMD --only bashF --headline-only > $OUT/stats_bashF.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- $CMD --only bashF,ctr,verify --ctr-gib 4 --headline-only > $OUT/pmc_$c.log 2>&1
done
# SQ passes over every workload of the bench (verification incl. one signer / keyed, signing, the fused kernel, the ragged hashes):
# pass 1 carries what valu_busy needs (SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 SEs x 1024 SIMDs)), pass 2 the LDS side
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -o bench -- $CMD --ctr-gib 4 --only bashF,ctr,verify,sign,mixed,ragged > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -o bench -- $CMD --ctr-gib 4 --only bashF,ctr,verify,sign,mixed,ragged > $OUT/pmc_sq2.log 2>&1
# condensed on the box (gpurun copies at most 64 MiB back; the raw per-launch CSVs of a full bench are more): the summaries land in
# gpurun_out/prof_summary/, to be copied into profiles/ here
TAG=${1:-r05}
PROF_DST=$R/gpurun_out/prof_summary python tools/summarize_prof.py $TAG > $OUT/summarize.log 2>&1; tail -3 $OUT/summarize.log
du -sh $OUT
rm -rf $OUT
