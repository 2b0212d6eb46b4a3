cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -k "video" 2>&1 | tail -8
