cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "peaked" 2>&1 | grep "peaked\]\|passed\|failed\|Error" | tail -40
