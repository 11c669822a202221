export TMPDIR=/tmp
for v in 0 1 2 0 1 2; do echo -n "HYP_POTRF_TIMING_ONLY=$v: "; HYP_POTRF_TIMING_ONLY=$v timeout 300 python tools/bench_potrf.py 5000 2>&1 | tail -1; done
echo -n "no look-ahead (one queue): "; HYP_POTRF_LOOKAHEAD=0 timeout 300 python tools/bench_potrf.py 5000 2>&1 | tail -1
