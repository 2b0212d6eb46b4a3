#!/bin/bash
# round 5: small batches through the stream-ordered entry, launches on the caller's stream (default) against the event-fenced path (VP_CALLER_STREAM=0), same box
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-host-path --no-clock"
one() { "$@" 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for cfg in "--variant l --dataset coco_25 --batch 8" "--variant b --batch 1"; do
  echo -n "caller-stream  $cfg: "; one timeout 30 $B $cfg
  echo -n "event-fenced   $cfg: "; VP_CALLER_STREAM=0 one timeout 30 $B $cfg
done > gpurun_out/callerstream_r5.txt 2>&1
cat gpurun_out/callerstream_r5.txt
