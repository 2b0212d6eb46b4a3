#!/bin/bash
# round 5, second session: (1) small batches in situ with the deep-ring tile configurations (measurement build), (2) the bench line, (3) the head-dim-80 / fused qkv + attention tests on the product build (HEAD of the first session)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R='default=;wide19=QKV:19:0,FC1:19:0;proj21=PROJ:21:0;fc2_22=FC2:22:0;fc2_23=FC2:23:0'
{
timeout 200 python tools/small_sweep.py --cases l:coco_25:8
timeout 120 python tools/small_sweep.py --cases l:coco_25:4,l:coco_25:16 --sets "$R"
timeout 120 python tools/small_sweep.py --cases b:coco:8,h:wholebody:4 --sets "$R"
} > gpurun_out/small_sweep_r5.txt 2>&1
tail -5 gpurun_out/small_sweep_r5.txt
timeout 240 python bench.py > gpurun_out/bench_r5b.json 2> gpurun_out/bench_r5b.err
tail -c 600 gpurun_out/bench_r5b.json
timeout 300 python -m pytest tests/test_gpu_api.py tests/test_gpu_ops.py -m gpu -x -q -k "head_dim_80 or fused_qkv or smoke or small" > gpurun_out/gputest_r5b.txt 2>&1
tail -4 gpurun_out/gputest_r5b.txt
