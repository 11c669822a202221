export TMPDIR=/tmp
python -m pytest tests/test_hip_cones.py tests/test_hip_solver.py -q -x -k "complex_hypograph or hyporootdettri_complex or hypoperlogdettri_complex" 2>&1 | tail -25
