cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm_cfgs.py tests/test_gpu_parity.py -x -q -k "configurations or patch_embed or final_conv or baseline" 2>&1 | tail -15
