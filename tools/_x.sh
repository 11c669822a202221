#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_switches.py -m gpu -q -x -k "certificate" 2>&1 | tail -5
for v in "" "HYP_WSOS_CERT=0"; do for c in 5p 5d; do echo "== $c $v"; env $v HYP_TRIAL_DBG=1 python bench.py --config $c 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['steps'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()}, d['roofline']['per_step']['bunch_kaufman_factorizations'], d['search_trials_per_step'])"; grep -c "early reject (infeasible)" /tmp/err.txt; grep -c "\] infeasible" /tmp/err.txt; done; done
