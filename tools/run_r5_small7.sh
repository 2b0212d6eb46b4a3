#!/bin/bash
# round 5, second session, call 7: PRODUCT build with the small-batch rule (Cfg31 / Cfg30 / Cfg12): every small configuration in situ, then the full GPU suite, then the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "== product build, default rule"
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 300 python tools/small_sweep.py --iters 100 --sets 'default=' --cases l:coco_25:8,l:coco_25:4,l:coco_25:2,l:coco_25:1,b:coco:8,b:coco:4,b:coco:2,b:coco:1,h:wholebody:8,h:wholebody:4,h:wholebody:1,s:coco:8,s:coco:1,l:coco_25:16,b:coco:16
} > gpurun_out/small_sweep7_r5.txt 2>&1
tail -2 gpurun_out/small_sweep7_r5.txt
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r5d.txt 2>&1
grep -E "passed|failed|error" gpurun_out/gputest_r5d.txt | tail -3
timeout 240 python bench.py > gpurun_out/bench_r5c.json 2> gpurun_out/bench_r5c.err
head -c 300 gpurun_out/bench_r5c.json
