#!/bin/bash
# round 3 closing run: full GPU suite + the default bench (its JSON line -> profiles/bench_r3.json) + smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3_pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_pytest_final.log
timeout 600 python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"; wc -l gpurun_out/r3_bench_final.json; head -c 900 gpurun_out/r3_bench_final.json; echo
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
