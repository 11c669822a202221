#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
run() { env $1 $2 $3 timeout 600 python bench.py --steps 60 --warmup 5 --cpu-iters 0 2>/tmp/e.txt | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); p = d['phases_ms_per_step']
print('$1 $2 $3', round(d['ms_per_step'], 3), 'sqrt', round(p['sqrt_hess_prod'], 3), 'syrk', round(p['syrk'], 3), 'chol', round(p['cholesky'], 3), 'upd', round(p['update_lhs'], 3), 'dir', round(p['get_directions'], 3), 'search', round(p['search'], 3))"; grep "^\[warm\]" /tmp/e.txt | head -1; }
run HYP_WARM=0 A=1 B=1
run HYP_WARM=150 HYP_WARM_WGS=256 HYP_WARM_PRIO=0
run HYP_WARM=150 HYP_WARM_WGS=64 HYP_WARM_PRIO=0
run HYP_WARM=0 A=1 B=1
run HYP_WARM=200 HYP_WARM_WGS=256 HYP_WARM_PRIO=0
