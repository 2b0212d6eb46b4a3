#!/bin/bash
# round 5, second session, call 6 (measurement build): 32 x 64 tiles with two k-blocks per barrier (Cfg31 6-stage 2 blocks / CU, Cfg32 8-stage 1 block / CU) against Cfg30 at 1-2 crops
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { timeout 200 python tools/small_sweep.py --iters 80 --cases "$1" --sets "$2"; }
ALL() { echo "QKV:$1:0,FC1:$1:0,PROJ:$1:0,FC2:$1:0"; }
{
run l:coco_25:1,b:coco:1,h:wholebody:1,s:coco:1,l:coco_25:2,b:coco:2 "default=;k30=$(ALL 30);k31=$(ALL 31);k32=$(ALL 32);mix=QKV:31:0,FC1:31:0,PROJ:32:0,FC2:32:0;default_b="
run l:coco_25:4,b:coco:4 'default=;k30=PROJ:30:0,FC2:30:0;k31=PROJ:31:0,FC2:31:0;k32=PROJ:32:0,FC2:32:0'
} > gpurun_out/small_sweep6_r5.txt 2>&1
tail -3 gpurun_out/small_sweep6_r5.txt
