cd $GRAFT_REPO_ROOT
timeout 600 python bench.py 2>&1 | tail -3
