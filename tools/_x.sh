export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_distributed.py -q -m gpu -x -k "row_groups_as_schur" 2>&1 | tail -8
