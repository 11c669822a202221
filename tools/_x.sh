export TMPDIR=/tmp
for f in 0 1 2 0 1 2; do echo "HYP_POTRF_EXPERIMENT=$f: $(HYP_POTRF_EXPERIMENT=$f timeout 300 python tools/bench_potrf.py 2>&1 | head -1)"; done
