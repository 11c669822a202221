#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -8 gpurun_out/pytest_gpu.log
