#!/bin/bash
# rocprofv3 --kernel-trace --stats of BASELINE configs[2]: ViTPose-H / wholebody, batch 128
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD
mkdir -p gpurun_out; rm -rf gpurun_out/prof_h
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_h/stats -o trace -- python $ROOT/bench.py --variant h --dataset wholebody --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-clock > $ROOT/gpurun_out/prof_h.log 2>&1
cd $ROOT
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --variant h --dataset wholebody --batch 128 --steps 5 --warmup 2 (BASELINE configs[2]), round 3 final binaries"; python tools/summarize_profile.py gpurun_out/prof_h 2>/dev/null | head -20; grep '^{' gpurun_out/prof_h.log | tail -1 | cut -c1-400; } > gpurun_out/h_rocprofv3.txt
rm -rf gpurun_out/prof_h
cat gpurun_out/h_rocprofv3.txt
