export TMPDIR=/tmp
run() { echo -n "$* : "; env "$@" timeout 600 python bench.py --steps 150 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_step']; print(round(d['ms_per_step'],3), 'shp %.2f syrk %.3f chol %.2f lhs %.2f dirs %.2f search %.2f' % (p['sqrt_hess_prod'],p['syrk'],p['cholesky'],p['update_lhs'],p['get_directions'],p['search']))"; }
run HYP_UPLOAD_AFTER=0
run HYP_UPLOAD_AFTER=1
run HYP_UPLOAD_AFTER=0
run HYP_UPLOAD_AFTER=1
run HYP_UPLOAD_AFTER=0
run HYP_UPLOAD_AFTER=1
timeout 600 python -m pytest tests/test_hip_switches.py tests/test_hip_solver.py -q -m gpu -x 2>&1 | tail -2
