cd $GRAFT_REPO_ROOT
python tools/gemm8_check.py --reps 1 2>&1 | grep -v "^#"
VP_G8_DEFERRED=1 python tools/determinism_check.py b 256 1 2>&1 | tail -2
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | cut -c1-120
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
