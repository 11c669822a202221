#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_switches.py tests/test_hip_distributed.py -q -x -m gpu 2>&1 | tail -3
run() { env $1 HYP_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 2w --no-secondary --steps 40 --warmup 3 2>/tmp/e.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); p = d['phases_ms_per_step']
print('$1', round(d['ms_per_step'], 3), 'upd', round(p['update_lhs'], 3), 'dir', round(p['get_directions'], 3), 'search', round(p['search'], 3), d.get('restarts_in_timed_region'), [(k, v) for k, v in d.items() if 'coll' in k][:2])" || tail -5 /tmp/e.txt; }
run A=1
run HYP_SEARCH_SCREEN_DIST=0
run A=1
