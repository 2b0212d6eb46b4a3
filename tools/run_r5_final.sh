#!/bin/bash
# round 5, second session, final call: PRODUCT build (small / mid batch rule: Cfg31 / Cfg30 / Cfg12 / Cfg15): the batch ladder in situ, the other-configuration bench lines that changed,
# the full GPU suite, the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5f
{
echo "== product build, shipped rule"
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 300 python tools/small_sweep.py --iters 100 --sets 'default=' --cases l:coco_25:1,l:coco_25:2,l:coco_25:4,l:coco_25:8,l:coco_25:12,l:coco_25:16,l:coco_25:24,b:coco:1,b:coco:4,b:coco:8,b:coco:16,b:coco:24,b:coco:32,h:wholebody:1,h:wholebody:4,h:wholebody:8,h:wholebody:12,s:coco:1,s:coco:8
} > gpurun_out/small_sweep8_r5.txt 2>&1
tail -1 gpurun_out/small_sweep8_r5.txt
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-host-path --no-clock"
timeout 120 $B --variant l --dataset coco_25 --batch 8 > gpurun_out/r5f/l_coco25_8.json 2> gpurun_out/r5f/l_coco25_8.err
timeout 120 $B --variant l --dataset coco_25 --batch 1 > gpurun_out/r5f/l_coco25_1.json 2> gpurun_out/r5f/l_coco25_1.err
timeout 120 $B --variant s --batch 1 > gpurun_out/r5f/s_coco_1.json 2> gpurun_out/r5f/s_coco_1.err
timeout 120 $B --variant b --batch 1 > gpurun_out/r5f/b_coco_1.json 2> gpurun_out/r5f/b_coco_1.err
for f in gpurun_out/r5f/*.json; do head -c 160 $f; echo; done
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r5e.txt 2>&1
grep -E "passed|failed|error" gpurun_out/gputest_r5e.txt | tail -3
timeout 240 python bench.py > gpurun_out/bench_r5d.json 2> gpurun_out/bench_r5d.err
head -c 300 gpurun_out/bench_r5d.json
