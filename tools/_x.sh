export TMPDIR=/tmp
python -m pytest tests/test_hip_cones.py -m gpu -q -x -k "beyond_96" 2>&1 | tail -5
HYP_TS3=0 python -m pytest tests/test_hip_cones.py -m gpu -q -x -k "beyond_96" 2>&1 | tail -3
