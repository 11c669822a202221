#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_switches.py -q -x -m gpu -k "screened" 2>&1 | tail -4
for e in "A=1" "HYP_SCREEN_SKIP_LB=0" "A=1" "HYP_SCREEN_SKIP_LB=0"; do
  env $e timeout 600 python bench.py --steps 60 --warmup 5 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$e', round(d['ms_per_step'], 3), 'search', round(d['phases_ms_per_step']['search'], 3), 'trials', d['search_trials_per_step'], 'screens', d.get('search_screens_per_step'), 'rej', d.get('search_trials_screened_out_per_step'))"
done
