#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_switches.py -m gpu -q -x -k "prefetched" 2>&1 | tail -5
for v in "" "HYP_LHS_PREFETCH=0" "" "HYP_LHS_PREFETCH=0"; do echo "== bench $v"; env $v python bench.py --steps 110 --cpu-iters 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['kkt_solves_per_step'], d['roofline']['frac'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
