#!/bin/bash
# round 5, second session, call 8 (measurement build): 12-32 crops in situ, residual GEMMs (proj / fc2) on the candidates between the 64 x 64 regime and the 8-phase kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { timeout 200 python tools/small_sweep.py --iters 50 --cases "$1" --sets "$2"; }
S='default=;r15=PROJ:15:0,FC2:15:0;r11=PROJ:11:0,FC2:11:0;r1=PROJ:1:0,FC2:1:0;p12f9=PROJ:12:0,FC2:9:0;p14f14=PROJ:14:0,FC2:14:0;r22=PROJ:22:0,FC2:22:0'
{
run b:coco:12,b:coco:16,b:coco:24,b:coco:32,l:coco_25:12,l:coco_25:16,l:coco_25:24,h:wholebody:12 "$S"
} > gpurun_out/mid_sweep_r5.txt 2>&1
tail -2 gpurun_out/mid_sweep_r5.txt
