#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_hip_fullsize_trajectory.py tests/test_hip_trajectory.py tests/test_hip_switches.py tests/test_hip_fullsize_configs.py -m gpu -q -x 2>&1 | tail -8
