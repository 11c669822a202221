#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
for v in "HYP_LANES_MASK=1" "HYP_LANES_MASK=2" "HYP_LANES_MASK=4" "HYP_LANES_MASK=0"; do for c in 5p; do echo "== $c $v"; env HYP_LANES=6 $v python bench.py --config $c --cpu-iters 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['steps'], d['config']['final_status'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"; done; done
