mkdir -p gpurun_out/r6f
(timeout 600 python -m pytest tests/test_gpu_production_sizes.py -k "content and (l-coco or h-whole)" -q -s -p no:cacheprovider 2>&1 | grep "^\[content\|^    crop\|passed\|failed" > gpurun_out/r6f/content_lines.txt)
(timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_production_sizes.py -q -p no:cacheprovider -k "not content" > gpurun_out/r6f/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6f/tests.log)
timeout 300 python tools/depth_probe.py > gpurun_out/r6f/depth_l8.txt 2>&1
timeout 200 python tools/depth_probe.py --variant b --dataset coco --crops 1 > gpurun_out/r6f/depth_b1.txt 2>&1
W='new=;old=QKV:1:0,FC1:1:0;new_b=;old_b=QKV:1:0,FC1:1:0'
timeout 400 python tools/small_sweep.py --iters 60 --cases l:coco_25:8,l:coco_25:7,l:coco_25:6,h:wholebody:8,h:wholebody:7 --sets "$W" 2>&1 | cut -c1-150 > gpurun_out/r6f/wide192.txt
cat gpurun_out/r6f/content_lines.txt | cut -c1-600; tail -4 gpurun_out/r6f/tests.log; cat gpurun_out/r6f/depth_l8.txt gpurun_out/r6f/depth_b1.txt gpurun_out/r6f/wide192.txt
