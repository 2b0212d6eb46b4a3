#!/bin/bash
# round 5, second session: rocprofv3 kernel statistics of the 8-crop share of BASELINE configs[3] (ViTPose-L) and of a single ViTPose-B crop, eager launches (VP_GRAPH=0: every kernel a traced dispatch;
# a hipGraph replay under the tracer ran into the call limit in the first session)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/prof_small
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --no-live-events"
VP_GRAPH=0 timeout 150 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_small/l8 -o trace -- $B --variant l --dataset coco_25 --batch 8 > $ROOT/gpurun_out/prof_small/l8.log 2>&1
VP_GRAPH=0 timeout 150 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_small/b1 -o trace -- $B --variant b --batch 1 > $ROOT/gpurun_out/prof_small/b1.log 2>&1
find $ROOT/gpurun_out/prof_small -name "*.csv" -size +8M -delete
find $ROOT/gpurun_out/prof_small -name "*.db" -size +30M -delete
ls -R $ROOT/gpurun_out/prof_small | head -30
