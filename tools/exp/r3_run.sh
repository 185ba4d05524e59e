cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_jobs.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
