# scratch batch (rewritten per call)
export TMPDIR=/tmp
for v in 0 2 3 0 2 3; do HYP_SYRK_SPLIT=$v python tools/bench_syrk.py 5000 20100 20 | sed "s/^/split $v: /"; done
for v in 0 2 3; do HYP_SYRK_SPLIT=$v python tools/bench_syrk.py 5000 207360 3 | sed "s/^/split $v cfg4: /"; done
for v in 0 3 2 0 3 2; do HYP_SYRK_SPLIT=$v python bench.py --cpu-iters 0 --steps 120 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 split $v:', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['phases_ms_per_step']['syrk'],3))"; done
HYP_SYRK_SPLIT=3 python -m pytest tests -m gpu -q -x -k "fullsize_trajectory and cfg2 or schur_probe or test_hip_dense" 2>&1 | tail -3
