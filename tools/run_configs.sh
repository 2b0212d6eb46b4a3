#!/bin/bash
# the other BASELINE configurations on one GPU: one JSON line each -> gpurun_out/configs/ (summaries: profiles/bench_other_configs_rN.txt)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/configs; mkdir -p $OUT
B="python $ROOT/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-path --no-clock"
run() { tag=$1; shift; timeout 300 $B "$@" > $OUT/$tag.json 2> $OUT/$tag.err || echo "FAILED $tag" >> $OUT/failed.txt; }
run s_coco_256 --variant s
run h_wholebody_128 --variant h --dataset wholebody --batch 128
run l_coco25_64 --variant l --dataset coco_25 --batch 64
run l_coco25_8 --variant l --dataset coco_25 --batch 8
run b_ap10k_512_fp16 --dataset ap10k --batch 512
run b_ap10k_512_fp8 --dataset ap10k --batch 512 --dtype fp8
run b_coco_256_fp8 --dtype fp8
run b_coco_256_bf16 --dtype bf16
run b_coco_256_u8 --input u8
run l_coco25_64_fp8 --variant l --dataset coco_25 --batch 64 --dtype fp8
run h_wholebody_128_fp8 --variant h --dataset wholebody --batch 128 --dtype fp8
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j = json.load(open(f))
        r = j["roofline"]
        print(f"{os.path.basename(f)[:-5]:24s} {j['value']:10.1f} persons/s  {j['ms_per_step']:8.3f} ms/step  {j['dtype']:5s} dominant {r['kernel']:34s} {1e3 * r['avg_launch_ms']:8.1f} us  {r['achieved']:7.1f} TF/s = {r['frac']:.3f} of {r['peak']:.0f}")
    except Exception as e:
        print(f, 'ERR', e)
PY
