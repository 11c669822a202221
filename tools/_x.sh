export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dense.py -q -m gpu -k "potrf" -x 2>&1 | tail -5
for v in 1 0 1 0; do echo -n "HYP_POTRF_DIAGUPD=$v: "; HYP_POTRF_DIAGUPD=$v timeout 300 python tools/bench_potrf.py 5000 4845 2250 2>&1 | tail -3 | tr '\n' ' '; echo; done
