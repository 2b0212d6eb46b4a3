#!/bin/bash
# round 5, second session, call 2: small batches in situ, deep-ring / k-block-32 / pipelined-read tile configurations per family (measurement build); per-family us are independent columns
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
L8='default=;A=QKV:19:0,FC1:19:0,PROJ:12:0,FC2:27:0;B=QKV:20:0,FC1:20:0,PROJ:14:0,FC2:12:0;C=QKV:24:0,FC1:24:0,PROJ:27:0,FC2:21:0;D=QKV:25:0,FC1:25:0,PROJ:21:0;E=QKV:26:0,FC1:26:0,PROJ:12:0;F=QKV:15:0,FC1:15:0,PROJ:12:0'
S='default=;A=QKV:14:0,FC1:14:0,PROJ:12:0,FC2:27:0;B=QKV:12:0,FC1:12:0,PROJ:23:0,FC2:12:0;C=QKV:21:0,FC1:21:0,PROJ:21:0,FC2:21:0;D=QKV:27:0,FC1:27:0,PROJ:14:0;E=QKV:25:0,FC1:25:0,PROJ:27:0'
P='default=;p12=PROJ:12:0;p14=PROJ:14:0;p21=PROJ:21:0'
{
timeout 150 python tools/small_sweep.py --iters 60 --cases l:coco_25:8,b:coco:8,h:wholebody:8 --sets "$L8"
timeout 200 python tools/small_sweep.py --iters 60 --cases l:coco_25:4,l:coco_25:2,l:coco_25:1,b:coco:4,b:coco:1,h:wholebody:4,h:wholebody:1,s:coco:8 --sets "$S"
timeout 100 python tools/small_sweep.py --iters 60 --cases l:coco_25:12,l:coco_25:16,b:coco:16,h:wholebody:12 --sets "$P"
} > gpurun_out/small_sweep2_r5.txt 2>&1
tail -3 gpurun_out/small_sweep2_r5.txt
