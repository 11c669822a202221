#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_dense.py tests/test_hip_cones.py -q -x -m gpu 2>&1 | tail -4
for e in "HYP_TRTRI_COLS=0" "HYP_TRTRI_COLS=1" "HYP_TRTRI_COLS=0" "HYP_TRTRI_COLS=1"; do
  env $e timeout 600 python bench.py --steps 60 --warmup 5 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$e', round(d['ms_per_step'], 3), 'upd', round(d['phases_ms_per_step']['update_lhs'], 3), 'chol', round(d['phases_ms_per_step']['cholesky'], 3), 'dir', round(d['phases_ms_per_step']['get_directions'], 3), 'kkt', d['kkt_solves_per_step'])"
done
timeout 1500 python -m pytest tests/test_hip_solver.py tests/test_hip_trajectory.py tests/test_hip_fullsize_configs.py tests/test_hip_switches.py -q -x -m gpu 2>&1 | tail -4
