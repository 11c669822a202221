export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_solver.py tests/test_hip_switches.py tests/test_hip_trajectory.py -q -x 2>&1 | tail -4
