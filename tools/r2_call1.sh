#!/bin/bash
# round-2 GPU call 1: gemm8 identity + timing, end-to-end A/B, chunked batches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c1; O=gpurun_out/c1
timeout 600 python tools/gemm8_check.py --reps 3 --ablate > $O/gemm8_check_b256.txt 2>&1
timeout 300 python tools/gemm8_check.py --batch 64 --reps 2 --iters 5 > $O/gemm8_check_b64.txt 2>&1
timeout 300 python tools/gemm8_check.py --variant l --batch 128 --reps 1 --iters 4 > $O/gemm8_check_l128.txt 2>&1
timeout 300 python tools/gemm8_check.py --variant s --batch 256 --reps 1 --iters 4 > $O/gemm8_check_s256.txt 2>&1
for cfg in 0 7 6 1 3 4; do
  echo "== VP_GEMM8=$cfg" >> $O/bench_ab.txt
  VP_GEMM8=$cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown >> $O/bench_ab.txt 2>&1
done
for mb in 128 64; do
  echo "== max-batch $mb VP_GEMM8=0" >> $O/bench_chunk.txt
  VP_GEMM8=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --max-batch $mb >> $O/bench_chunk.txt 2>&1
  echo "== max-batch $mb VP_GEMM8=7" >> $O/bench_chunk.txt
  VP_GEMM8=7 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --max-batch $mb >> $O/bench_chunk.txt 2>&1
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cat $O/gemm8_check_b256.txt
grep -h "value\|==" $O/bench_ab.txt | cut -c1-200
