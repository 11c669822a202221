export TMPDIR=/tmp
python -m pytest tests/test_hip_cones.py tests/test_hip_solver.py -q -x -k "linmatrixineq" 2>&1 | tail -5
