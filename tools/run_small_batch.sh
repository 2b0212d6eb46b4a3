#!/bin/bash
# Small / mid batches on the PRODUCT build (round 5): the batch ladder in situ (tools/small_sweep.py: ms per step + us per launch of the encoder GEMM families + the kernels
# the rule picked), the bench lines of the 8-crop share of BASELINE configs[3] and of single crops, hipGraph replay against eager launches.  -> gpurun_out/small_batch/
# (profiles/small_batch_rN.txt hold the outputs of this script and of the exploratory sweeps that led to the rule; rocprofv3 of the same paths: tools/profile.sh small)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/small_batch; mkdir -p $OUT
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 300 python tools/small_sweep.py --iters 100 --sets 'default=' \
  --cases l:coco_25:1,l:coco_25:2,l:coco_25:4,l:coco_25:8,l:coco_25:12,l:coco_25:16,l:coco_25:24,b:coco:1,b:coco:4,b:coco:8,b:coco:16,b:coco:24,b:coco:32,h:wholebody:1,h:wholebody:4,h:wholebody:8,h:wholebody:12,s:coco:1,s:coco:8 > $OUT/ladder.txt 2>&1
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-host-path --no-clock"
one() { "$@" 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for cfg in "--variant l --dataset coco_25 --batch 8" "--variant l --dataset coco_25 --batch 1" "--variant b --batch 1" "--variant s --batch 1" "--variant b --batch 16"; do
  echo -n "graph  $cfg: "; one timeout 100 $B $cfg
  echo -n "eager  $cfg: "; VP_GRAPH=0 one timeout 100 $B $cfg
done > $OUT/bench_lines.txt 2>&1
# the stream-ordered entry: launches on the caller's stream (default at <= 16 crops) against the event-fenced path, same box
for cfg in "--variant l --dataset coco_25 --batch 8" "--variant b --batch 1"; do
  echo -n "caller-stream  $cfg: "; one timeout 60 $B $cfg
  echo -n "event-fenced   $cfg: "; VP_CALLER_STREAM=0 one timeout 60 $B $cfg
done > $OUT/callerstream.txt 2>&1
tail -3 $OUT/ladder.txt; cat $OUT/bench_lines.txt $OUT/callerstream.txt
