#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python bench.py --steps 60 --warmup 5 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); p = d['phases_ms_per_step']
print(round(d['ms_per_step'], 3), 'upd', round(p['update_lhs'], 3), 'chol', round(p['cholesky'], 3), 'dir', round(p['get_directions'], 3))"; done
timeout 900 python -m pytest tests/test_hip_dense.py -q -x -m gpu -k "posv or potrf or trsv or solve" 2>&1 | tail -2
