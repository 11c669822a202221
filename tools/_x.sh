#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
for v in "HYP_SYRK_EDGE_SIDE=1" "HYP_SYRK_EDGE_SIDE=0" "HYP_SYRK_EDGE_SIDE=1" "HYP_SYRK_EDGE_SIDE=0"; do echo "== $v"; env $v python bench.py --cpu-iters 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['steps'], round(d['roofline']['frac'],4), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()}, round(d['kkt_solves_per_step'],2))"; done
for v in "HYP_SYRK_EDGE_SIDE=1" "HYP_SYRK_EDGE_SIDE=0"; do echo "== cfg4 $v"; env $v python bench.py --config 4 --cpu-iters 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['steps'], round(d['roofline']['frac'],4), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"; done
timeout 1500 python -m pytest tests/test_hip_fullsize.py tests/test_hip_dense.py tests/test_hip_fullsize_configs.py -m gpu -q -x 2>&1 | tail -3
