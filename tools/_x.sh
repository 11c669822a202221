#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_solver.py -q -x -m gpu -k "resident or native_search" 2>&1 | tail -15
