mkdir -p gpurun_out/r6c
SETS='default=;f2_31=|fc2:2:31;f4_31=|fc2:4:31;f2_12=|fc2:2:12;f4_12=|fc2:4:12;f2_15=|fc2:2:15;f4_15=|fc2:4:15;f2_1=|fc2:2:1;f4_1=|fc2:4:1'
for m in b:coco s:coco l:coco_25 h:wholebody; do
  cases=""; for n in 1 2 4 6 8 12; do cases="$cases,$m:$n"; done
  timeout 900 python tools/small_sweep.py --iters 50 --cases ${cases#,} --sets "$SETS" 2>&1 | cut -c1-96 > gpurun_out/r6c/grid_${m%%:*}.txt
done
cat gpurun_out/r6c/grid_*.txt
