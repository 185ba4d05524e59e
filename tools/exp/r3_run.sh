cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_jobs.py tests/test_examples.py tests/test_mdn.py tests/test_export.py tests/test_pocket.py tests/test_pose_init.py tests/test_real_complex.py -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
