export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_solver.py tests/test_hip_trajectory.py tests/test_hip_distributed.py tests/test_hip_fullsize.py tests/test_hip_fullsize_configs.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
