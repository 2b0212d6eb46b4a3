#!/bin/bash
# round 5, second session, call 4: (a) PRODUCT build with the <= 512-tile 4-stage rule at small batches, (b) tile order at 8 crops (measurement build, three repeats), (c) model-level GPU tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "== product build, default rule"
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 200 python tools/small_sweep.py --iters 100 --sets 'default=' --cases l:coco_25:8,l:coco_25:4,l:coco_25:2,l:coco_25:1,b:coco:8,b:coco:4,b:coco:1,h:wholebody:8,h:wholebody:4,h:wholebody:1,s:coco:8,s:coco:1
echo "== measurement build: tile order at 8 crops (group_m of the wide / residual GEMMs)"
G='g0=QKV:1:0,FC1:1:0,PROJ:12:0,FC2:12:0;g4_8=QKV:1:4,FC1:1:4,PROJ:12:8,FC2:12:8;g2_4=QKV:1:2,FC1:1:2,PROJ:12:4,FC2:12:4;g6_12=QKV:1:6,FC1:1:6,PROJ:12:12,FC2:12:12;g0b=QKV:1:0,FC1:1:0,PROJ:12:0,FC2:12:0;g4_8b=QKV:1:4,FC1:1:4,PROJ:12:8,FC2:12:8;g0c=QKV:1:0,FC1:1:0,PROJ:12:0,FC2:12:0;g4_8c=QKV:1:4,FC1:1:4,PROJ:12:8,FC2:12:8'
timeout 200 python tools/small_sweep.py --iters 100 --cases l:coco_25:8,h:wholebody:8 --sets "$G"
} > gpurun_out/small_sweep4_r5.txt 2>&1
tail -3 gpurun_out/small_sweep4_r5.txt
timeout 400 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/gputest_r5c.txt 2>&1
tail -4 gpurun_out/gputest_r5c.txt
