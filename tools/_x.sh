#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_switches.py -q -x -m gpu -k "screened" 2>&1 | tail -8
run() { env $1 timeout 900 python bench.py --config 4 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); p = d['phases_ms_per_step']
print('$1', round(d['ms_per_step'], 3), 'upd', round(p['update_lhs'], 3), 'dir', round(p['get_directions'], 3), 'search', round(p['search'], 3), 'trials', d['search_trials_per_step'], 'screens', d.get('search_screens_per_step'), 'rej', d.get('search_trials_screened_out_per_step'), 'restarts', d.get('restarts_in_timed_region'))"; }
run A=1
run HYP_SEARCH_SCREEN_RUN=0
run A=1
