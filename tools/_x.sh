#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
for v in "HYP_POTRF_AGG_EXPERIMENT=0" "HYP_POTRF_AGG_EXPERIMENT=1" "HYP_POTRF_AGG_EXPERIMENT=0" "HYP_POTRF_AGG_EXPERIMENT=1"; do echo "== $v"; env $v python tools/bench_potrf.py 5000 4845 2>&1 | tail -2; done
