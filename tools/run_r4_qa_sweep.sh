#!/bin/bash
# fused qkv + attention vs two launches over the batch size (ViTPose-B and -L): where does the fused kernel start to pay?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4l; mkdir -p $OUT
for cfg in "b coco 32" "b coco 44" "b coco 64" "b coco 86" "b coco 128" "b coco 192" "l coco_25 16" "l coco_25 32" "l coco_25 64"; do
  set -- $cfg
  for f in 0 1; do
    VP_FUSE_QKV_ATTN=$f VP_QA_MIN_TILES=8 timeout 200 python $ROOT/bench.py --variant $1 --dataset $2 --batch $3 --steps 60 --warmup 10 --no-cpu-baseline --no-host-path --no-clock > $OUT/s_$1_$3_$f.json 2>/dev/null
    python - <<PY
import json
j=json.load(open("$OUT/s_$1_$3_$f.json"))
print("$1 n=$3 fuse=$f tiles=%d: %9.1f persons/s %8.3f ms  qkv family %s %s us" % (int("$3")//2*(12 if "$1"=="b" else 16), j["value"], j["ms_per_step"], j["encoder_gemms"]["gemm_qkv"]["kernel"][:28], j["encoder_gemms"]["gemm_qkv"]["avg_launch_us"]))
PY
  done
done
