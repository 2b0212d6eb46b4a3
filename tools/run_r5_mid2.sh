#!/bin/bash
# round 5, second session, call 9 (measurement build): the 128(m) x 64(n) 3-stage tile (Cfg15) for the WIDE GEMMs where the 64 x 64 regime has more than 512 tiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { timeout 200 python tools/small_sweep.py --iters 50 --cases "$1" --sets "$2"; }
{
run l:coco_25:4,h:wholebody:4,s:coco:8,b:coco:8,s:coco:16,b:coco:6,l:coco_25:6 'default=;w15=QKV:15:0,FC1:15:0;q15=QKV:15:0;f15=FC1:15:0;default_b='
} > gpurun_out/mid_sweep2_r5.txt 2>&1
tail -2 gpurun_out/mid_sweep2_r5.txt
