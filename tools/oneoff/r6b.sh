mkdir -p gpurun_out/r6b
(timeout 600 python -m pytest tests/test_gpu_gemm_cfgs.py -k split_k -x -q -s -p no:cacheprovider > gpurun_out/r6b/splitk_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6b/splitk_tests.log)
(timeout 600 python -m pytest tests/test_gpu_production_sizes.py -k content -q -s -p no:cacheprovider > gpurun_out/r6b/content_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6b/content_tests.log)
L8='default=;f2_1=|fc2:2:1;f4_1=|fc2:4:1;f2_20=|fc2:2:20;f4_20=|fc2:4:20;f4_11=|fc2:4:11;f8_20=|fc2:8:20;f2_15=|fc2:2:15;f4_15=|fc2:4:15;f2_12=|fc2:2:12;f4_12=|fc2:4:12;f2_30=|fc2:2:30;f4_30=|fc2:4:30;p2_12=|proj:2:12;p2_30=|proj:2:30;p2_15=|proj:2:15;default_b='
timeout 500 python tools/small_sweep.py --iters 60 --cases l:coco_25:8 --sets "$L8" > gpurun_out/r6b/sweep_l8.txt 2>&1
B1='default=;f2_31=|fc2:2:31;f4_31=|fc2:4:31;f8_31=|fc2:8:31;f4_30=|fc2:4:30;f8_30=|fc2:8:30;f4_12=|fc2:4:12;p2_31=|proj:2:31;p3_31=|proj:3:31;f4p2=|fc2:4:31,proj:2:31;default_b='
timeout 300 python tools/small_sweep.py --iters 100 --cases b:coco:1,l:coco_25:1 --sets "$B1" > gpurun_out/r6b/sweep_1.txt 2>&1
H8='default=;f2_20=|fc2:2:20;f4_20=|fc2:4:20;f4_1=|fc2:4:1;f5_20=|fc2:5:20;f4_15=|fc2:4:15'
timeout 300 python tools/small_sweep.py --iters 40 --cases h:wholebody:8,l:coco_25:4,b:coco:8,l:coco_25:16 --sets "$H8" > gpurun_out/r6b/sweep_misc.txt 2>&1
tail -5 gpurun_out/r6b/splitk_tests.log; tail -12 gpurun_out/r6b/content_tests.log; cat gpurun_out/r6b/sweep_l8.txt gpurun_out/r6b/sweep_1.txt gpurun_out/r6b/sweep_misc.txt
