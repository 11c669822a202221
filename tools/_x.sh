export TMPDIR=/tmp
for rep in 1 2; do
for s in 0 5 6 7 4 8; do echo -n "HYP_SYRK_S=$s: "; HYP_SYRK_S=$s timeout 300 python tools/bench_syrk.py 5000 20100 20 2>&1 | tail -1; done
done
for s in 0 5 7; do echo "== bench HYP_SYRK_S=$s"; HYP_SYRK_S=$s timeout 600 python bench.py --steps 100 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_step']; print(d['ms_per_step'], 'shp %.2f syrk %.2f chol %.2f lhs %.2f dirs %.2f' % (p['sqrt_hess_prod'],p['syrk'],p['cholesky'],p['update_lhs'],p['get_directions']))"; done
