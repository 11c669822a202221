#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for v in 0 5000 3072 2560 0 3072; do echo "T_EARLY=$v: $(HYP_POTRF_T_EARLY=$v timeout 300 python tools/bench_potrf.py 2>&1 | head -2 | tr '\n' ' ')"; done
HYP_POTRF_T_EARLY=5000 timeout 900 python -m pytest tests/test_hip_dense.py -q -x -m gpu -k "potrf or posv" 2>&1 | tail -2
