export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_dense.py -m gpu -q --tb=short 2>&1 | tail -12
echo "--- refine=0 for comparison"
HYP_TRSM_REFINE=0 timeout 600 python -m pytest tests/test_hip_dense.py -m gpu -q --tb=line -k posv_multi 2>&1 | grep -v "^/tmp" | tail -14
