#!/bin/bash
# rocprofv3 --kernel-trace --stats of the small-batch operating point (ViTPose-L / coco_25, 8 crops: one GPU's share of BASELINE configs[3])
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD
mkdir -p gpurun_out; rm -rf gpurun_out/prof_small_f0 gpurun_out/prof_small_f8
cd /tmp && export TMPDIR=/tmp
for f in 0 8; do
VP_GRAPH=0 VP_FOLD_STATS=$f rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_small_f$f/stats -o trace -- python $ROOT/bench.py --variant l --dataset coco_25 --batch 8 --input u8 --steps 50 --warmup 5 --no-cpu-baseline --no-host-path --no-clock > $ROOT/gpurun_out/prof_small_f$f.log 2>&1
done
cd $ROOT
for f in 0 8; do echo "== VP_FOLD_STATS=$f (VP_GRAPH=0 so that every kernel is a traced dispatch)"; python tools/summarize_profile.py gpurun_out/prof_small_f$f 2>/dev/null | head -16; done > gpurun_out/small_rocprofv3.txt
rm -rf gpurun_out/prof_small_f0 gpurun_out/prof_small_f8
cat gpurun_out/small_rocprofv3.txt
