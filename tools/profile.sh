#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (run on the GPU box via gpurun).
#   tools/profile.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/{stats,pmc*}
#   tools/profile.sh small                      -> gpurun_out/prof_small/{l8,b1}: kernel statistics of the 8-crop share of BASELINE configs[3]
#                                                  (ViTPose-L) and of ONE ViTPose-B crop, eager launches (VP_GRAPH=0: every kernel a traced dispatch)
# Kernel-trace/stats and every PMC pass are separate runs (PMC is never combined with
# --sys-trace / hip / hsa trace domains).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$TAG" = small ]; then
  mkdir -p $ROOT/gpurun_out/prof_small
  cd /tmp && export TMPDIR=/tmp
  B="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --no-live-events"
  VP_GRAPH=0 timeout 150 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_small/l8 -o trace -- $B --variant l --dataset coco_25 --batch 8 > $ROOT/gpurun_out/prof_small/l8.log 2>&1
  VP_GRAPH=0 timeout 150 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_small/b1 -o trace -- $B --variant b --batch 1 > $ROOT/gpurun_out/prof_small/b1.log 2>&1
  find $ROOT/gpurun_out/prof_small -name "*.csv" -size +8M -delete
  find $ROOT/gpurun_out/prof_small -name "*.db" -size +30M -delete
  ls -R $ROOT/gpurun_out/prof_small | head -30
  exit 0
fi
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-host-path --no-clock --steps 5 --warmup 2 $*"
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- $BENCH > $OUT/stats.log 2>&1
PMCB="python $ROOT/bench.py --no-cpu-baseline --no-host-path --no-clock --steps 1 --warmup 1 $*"
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o pmc -- $PMCB > $OUT/pmc_sq.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $OUT/pmc_tcc -o pmc -- $PMCB > $OUT/pmc_tcc.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $PMCB > $OUT/pmc_write.log 2>&1
# keep only the small CSVs
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -50
