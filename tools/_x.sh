export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_dense.py -q -x -m gpu -k "one_launch or two_contexts or superblock" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_switches.py -q -x -m gpu -k "one_launch" 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_hip_distributed.py -q -x -m gpu 2>&1 | tail -3
HYP_TRSV_OL_STATS=1 timeout 600 python bench.py --steps 30 --cpu-iters 0 2>&1 >/dev/null | grep "one-launch"
