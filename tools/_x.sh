#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_switches.py tests/test_hip_distributed.py tests/test_hip_solver.py -q -x -m gpu 2>&1 | tail -3
