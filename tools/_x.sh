#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_distributed.py -q -x -m gpu -k "one_equal_cone or runs_of_equal" 2>&1 | tail -5
