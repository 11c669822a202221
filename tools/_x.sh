export TMPDIR=/tmp
run() { echo -n "$* : "; env "$@" python bench.py --steps 150 --warmup 20 --cpu-iters 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_step']; print(round(d['ms_per_step'],3), 'syrk %.2f chol %.2f lhs %.2f dirs %.2f search %.2f' % (p['syrk'],p['cholesky'],p['update_lhs'],p['get_directions'],p['search']))"; }
for i in 1 2 3; do run HYP_SYNC_SPIN=0; run HYP_SYNC_SPIN=1; run HYP_SYNC_SPIN=0 HSA_ENABLE_INTERRUPT=0; run HYP_SYNC_SPIN=1 HSA_ENABLE_INTERRUPT=0; done
for c in 3b 5p; do for v in 0 1; do echo -n "config $c HYP_SYNC_SPIN=$v : "; HYP_SYNC_SPIN=$v timeout 600 python bench.py --config $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; done; done
