export TMPDIR=/tmp
python -m pytest tests/test_hip_distributed.py -q -x 2>&1 | tail -3
