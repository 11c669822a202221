export TMPDIR=/tmp
python tools/bench_syrk.py 5000 20100 8 | tail -1
python tools/bench_potrf.py 5000 4845 1000 | tail -3
python tools/bench_trsv.py 5000 | tail -1
python -m pytest tests/test_hip_dense.py -q -x 2>&1 | tail -2
