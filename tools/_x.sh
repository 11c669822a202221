#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/_dbg_res.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_switches.py -q -x -m gpu -k "screened" 2>&1 | tail -8
for e in "A=1" "HYP_SEARCH_RESIDENT=0" "HYP_SEARCH_SCREEN=0" "A=1"; do
  env $e timeout 600 python bench.py --steps 60 --warmup 5 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$e', round(d['ms_per_step'], 3), 'search', round(d['phases_ms_per_step']['search'], 3), 'trials', d['search_trials_per_step'], 'screens', d.get('search_screens_per_step'), 'rej', d.get('search_trials_screened_out_per_step'))"
done
timeout 1200 python -m pytest tests/test_hip_trajectory.py tests/test_hip_solver.py tests/test_hip_fullsize_configs.py -q -x -m gpu 2>&1 | tail -3
