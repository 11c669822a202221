#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
for f in 0 8 16 32 48 64 96; do echo "== HYP_POTRF_FREE_CUS=$f"; HYP_POTRF_FREE_CUS=$f python tools/bench_potrf.py 5000 4845 2250 2>&1 | tail -3; done
