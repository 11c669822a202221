#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_bunchkaufman.py -m gpu -q -x 2>&1 | tail -3
for v in "HYP_POTRF_LOOKAHEAD=1" "HYP_POTRF_LOOKAHEAD=0" "HYP_POTRF_LA_MIN=2560" "HYP_POTRF_LA_MIN=3584" "HYP_POTRF_LA_MIN=768" "HYP_POTRF_TRAIL_TILE=128" "HYP_POTRF_TRAIL_TILE=0"; do echo "== $v"; env $v python tools/bench_potrf.py 5000 2>&1 | tail -1; done
