cd $GRAFT_REPO_ROOT
python tools/gemm_selfcheck.py 384 16 2>&1 | grep -v "noln"
python tools/gemm_selfcheck.py 768 8 2>&1 | grep "cfg9 vs cfg9\|cfg1 vs cfg1"
python tools/determinism_check.py s 16 3
python tools/determinism_check.py b 8 2
python tools/gemm8_check.py --reps 1 --no-bench
