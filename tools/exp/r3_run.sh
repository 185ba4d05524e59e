mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
timeout 900 python tests/golden/make_oracle_fixtures.py gpurun_out/r3 2 2>&1 | grep -v Warning | tail -2
cp gpurun_out/r3/cfg2_traj.npz tests/golden/
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cfg_shape" 2>&1 | tail -3
