#!/bin/bash
# closing run of round 3: full GPU suite, smoke, the default bench line, the other BASELINE configs and the frame stream on the final binaries
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3_pytest_final.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3_pytest_final.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"; head -c 200 gpurun_out/r3_bench_final.json; echo
{
echo "# python bench.py <cfg> --steps 30 --warmup 5 --no-cpu-baseline --no-host-path, one MI355X, round 3 (product library, final binaries)"
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
timeout 300 python tools/stream_bench.py --frames 100 > gpurun_out/r3_stream.txt 2>&1; grep '^{' gpurun_out/r3_stream.txt | tail -1 | cut -c1-330
timeout 300 python tools/stream_bench.py --frames 100 --persons 8 > gpurun_out/r3_stream8.txt 2>&1; grep '^{' gpurun_out/r3_stream8.txt | tail -1 | cut -c1-330
