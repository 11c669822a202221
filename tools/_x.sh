export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_cones.py -q -x -m gpu -k "on_chip or one_wavefront or sides_beyond" 2>&1 | tail -3
cd /tmp
for side in 96 112 128; do for tm in 6 9; do rm -rf /tmp/pp; HYP_TS4_TMIN=$tm rocprofv3 --kernel-trace --stats -d /tmp/pp -o b -- python $GRAFT_REPO_ROOT/tools/bench_psd_ts.py $side 5000 3 > /dev/null 2>&1; echo "side $side HYP_TS4_TMIN=$tm"; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pp -name "*.db" | head -1) 2>/dev/null | grep -i "psd_ts" | cut -c1-140; done; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/ts4_small_t.txt
