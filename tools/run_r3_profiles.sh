#!/bin/bash
# round 3 evidence: rocprofv3 stats + PMC passes of the default bench command, the other BASELINE configs, the frame stream, --force-dist
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r3
bash tools/profile.sh r3 > gpurun_out/prof_r3.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r3 gpurun_out/pmc_r3.json > gpurun_out/r3_rocprofv3.txt 2>&1
head -30 gpurun_out/r3_rocprofv3.txt
rm -rf gpurun_out/prof_r3/*/  # keep only the summaries (the .db files are large)
{
echo "# python bench.py <cfg> --steps 30 --warmup 5 --no-cpu-baseline --no-host-path, one MI355X, round 3 (product library)"
for cfg in "--variant s --dataset coco --batch 256" "--variant h --dataset wholebody --batch 128" "--variant l --dataset coco_25 --batch 64" \
           "--variant l --dataset coco_25 --batch 8 --input u8" "--variant b --dataset ap10k --batch 512" "--variant b --dataset coco --batch 256 --dtype bf16" \
           "--variant b --dataset coco --batch 256 --input u8"; do
  echo "== $cfg"
  timeout 300 python bench.py $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'],'persons/s',d['ms_per_step'],'ms/step',d['model_tflops'],'TF model; dominant',r['kernel'],':',r['what'],'avg launch us',round(r['avg_launch_ms']*1e3,1),'frac',r['frac'])
"
done
} > gpurun_out/r3_other_configs.txt 2>&1
cat gpurun_out/r3_other_configs.txt
timeout 300 python tools/stream_bench.py --frames 100 > gpurun_out/r3_stream.txt 2>&1; tail -3 gpurun_out/r3_stream.txt
timeout 300 python bench.py --force-dist --strong --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-clock > gpurun_out/r3_force_dist.json 2> gpurun_out/r3_force_dist.err; echo "force-dist rc=$?"; head -c 600 gpurun_out/r3_force_dist.json; tail -3 gpurun_out/r3_force_dist.err
