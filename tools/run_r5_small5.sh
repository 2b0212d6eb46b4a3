#!/bin/bash
# round 5, second session, call 5 (measurement build): two k-blocks per barrier on the 64 x 64 tiles (PIPE 6: Cfg28 5-stage, Cfg29 4-stage, Cfg30 6-stage), and the tile order at 8-12 crops of the other models
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { timeout 150 python tools/small_sweep.py --iters 80 --cases "$1" --sets "$2"; }
ALL() { echo "QKV:$1:0,FC1:$1:0,PROJ:$1:0,FC2:$1:0"; }
{
run l:coco_25:1,b:coco:1,h:wholebody:1,s:coco:1,l:coco_25:2 "default=;k28=$(ALL 28);k29=$(ALL 29);k30=$(ALL 30);default_b="
run l:coco_25:8,l:coco_25:4,h:wholebody:4,b:coco:4,b:coco:8,h:wholebody:8 'default=;k28=PROJ:28:0,FC2:28:0;k29=PROJ:29:0,FC2:29:0;k30=PROJ:30:0,FC2:30:0;default_b='
run b:coco:8 'g0=QKV:9:0,FC1:1:0,PROJ:12:0,FC2:12:0;g=QKV:9:8,FC1:1:4,PROJ:12:8,FC2:12:8;g0b=QKV:9:0,FC1:1:0,PROJ:12:0,FC2:12:0;gb=QKV:9:8,FC1:1:4,PROJ:12:8,FC2:12:8'
run s:coco:8 'g0=QKV:12:0,FC1:9:0,PROJ:12:0,FC2:12:0;g=QKV:12:8,FC1:9:8,PROJ:12:8,FC2:12:8;g0b=QKV:12:0,FC1:9:0,PROJ:12:0,FC2:12:0;gb=QKV:12:8,FC1:9:8,PROJ:12:8,FC2:12:8'
run l:coco_25:12 'g0=QKV:1:0,PROJ:9:0,FC2:12:0;g=QKV:1:4,PROJ:9:8,FC2:12:8;g0b=QKV:1:0,PROJ:9:0,FC2:12:0;gb=QKV:1:4,PROJ:9:8,FC2:12:8'
} > gpurun_out/small_sweep5_r5.txt 2>&1
tail -3 gpurun_out/small_sweep5_r5.txt
