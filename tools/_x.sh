export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_switches.py -q -m gpu -x -k "two_column_groups" 2>&1 | tail -15
