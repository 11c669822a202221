export TMPDIR=/tmp
python -m pytest tests/test_hip_dense.py -m gpu -q -x -k "gemv" 2>&1 | tail -3
rm -rf /tmp/pb; rocprofv3 --kernel-trace -d /tmp/pb -o b -- python bench.py --steps 20 --cpu-iters 0 > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/pb -name "*.db" | head -1) 2>/dev/null | grep -i "gemv_both\|reduce\|name" | cut -c1-150
python -m pytest tests/test_hip_solver.py tests/test_hip_trajectory.py tests/test_hip_distributed.py tests/test_hip_fullsize.py -m gpu -q -x 2>&1 | tail -3
