ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/prof_r5b; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- python $ROOT/bench.py --no-cpu-baseline --no-host-path --no-clock --steps 12 --warmup 3 > $OUT/stats.log 2>&1
find $OUT -name "*.csv" -size +8M -delete; find $OUT -name "*.db" -size +40M -delete; grep -o '"value": [0-9.]*' $OUT/stats.log | head -1; ls $OUT/stats
