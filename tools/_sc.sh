timeout 100 python tools/gemm8_check.py 2>&1 | grep -E "compare|v17|cfg11"
timeout 100 python tools/gemm8_timeline.py --resid 2>&1 | tail -8 | cut -c1-260
