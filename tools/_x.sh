export TMPDIR=/tmp
python -m pytest tests/test_hip_dense.py -m gpu -q -x -k "syrk" 2>&1 | tail -3
for cb in 8 4; do
rm -rf /tmp/p4; HYP_SYRK_EDGE_CB=$cb rocprofv3 --kernel-trace -d /tmp/p4 -o b -- python bench.py --config 4 --steps 4 --warmup 1 --cpu-iters 0 > /dev/null 2>&1
echo cb $cb; python tools/rocpd_stats.py $(find /tmp/p4 -name "*.db" | head -1) 2>/dev/null | grep "gemm_f64_kernel<true, 4, 1>\|syrk_edge\|splitk" | cut -c1-160
done
