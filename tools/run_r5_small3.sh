#!/bin/bash
# round 5, second session, call 3: small batches in situ: 4-stage 64 x 64 tiles up to 512 tiles (the candidate rule) without / with the m-fastest tile order over ALL m-tiles
# (group_m >= tiles_m: every XCD then owns a range of n-tiles = every HBM-cold weight tile is fetched by one XCD instead of eight)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { timeout 120 python tools/small_sweep.py --iters 60 --cases "$1" --sets "$2"; }
{
run l:coco_25:8,h:wholebody:8 'default=;g0=QKV:1:0,FC1:1:0,PROJ:12:0,FC2:12:0;g64=QKV:1:64,FC1:1:64,PROJ:12:64,FC2:12:64;g64w20=QKV:20:64,FC1:20:64,PROJ:12:64,FC2:12:64;g4=QKV:1:4,FC1:1:4,PROJ:12:8,FC2:12:8'
run b:coco:8 'default=;g0=QKV:9:0,FC1:1:0,PROJ:12:0,FC2:12:0;g64=QKV:9:64,FC1:1:64,PROJ:12:64,FC2:12:64;g64w20=QKV:20:64,FC1:20:64,PROJ:12:64,FC2:12:64'
run l:coco_25:4,h:wholebody:4 'default=;g0=QKV:9:0,FC1:9:0,PROJ:12:0,FC2:12:0;g64=QKV:9:64,FC1:9:64,PROJ:12:64,FC2:12:64;g64x3=QKV:14:64,FC1:14:64,PROJ:12:64,FC2:12:64'
run b:coco:4,s:coco:8 'default=;g0=QKV:12:0,FC1:9:0,PROJ:12:0,FC2:12:0;g64=QKV:12:64,FC1:9:64,PROJ:12:64,FC2:12:64;g64x3=QKV:12:64,FC1:14:64,PROJ:12:64,FC2:12:64'
run l:coco_25:1,l:coco_25:2,b:coco:1,h:wholebody:1,s:coco:1 'default=;g0=QKV:12:0,FC1:12:0,PROJ:12:0,FC2:12:0;g64=QKV:12:64,FC1:12:64,PROJ:12:64,FC2:12:64'
} > gpurun_out/small_sweep3_r5.txt 2>&1
tail -3 gpurun_out/small_sweep3_r5.txt
