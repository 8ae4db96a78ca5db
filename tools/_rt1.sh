cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_rowtile.py -m gpu -q -k "never_arrives" 2>&1 | tail -5 | cut -c1-250
