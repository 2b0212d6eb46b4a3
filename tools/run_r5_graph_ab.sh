#!/bin/bash
# round 5, second session: hipGraph replay against eager launches at small batches, PRODUCT build, same box (bench.py, 300 steps; and the un-ordered device entry of tools/small_sweep.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-host-path --no-clock"
one() { "$@" 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
{
for r in 1 2; do
for cfg in "--variant l --dataset coco_25 --batch 8" "--variant b --batch 1" "--variant l --dataset coco_25 --batch 1" "--variant b --batch 16"; do
  echo -n "graph  $cfg: "; one timeout 100 $B $cfg
  echo -n "eager  $cfg: "; VP_GRAPH=0 one timeout 100 $B $cfg
done; done
echo "== un-ordered device entry (tools/small_sweep.py, product build): graph"
VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 100 python tools/small_sweep.py --iters 200 --sets 'default=' --cases l:coco_25:8,b:coco:1 2>&1 | grep default
echo "== eager"
VP_GRAPH=0 VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 100 python tools/small_sweep.py --iters 200 --sets 'default=' --cases l:coco_25:8,b:coco:1 2>&1 | grep default
} > gpurun_out/graph_ab_r5.txt 2>&1
cat gpurun_out/graph_ab_r5.txt | cut -c1-120
