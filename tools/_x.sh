#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_cones.py -m gpu -q -x -k "on_chip" 2>&1 | tail -3
HYP_TS4_PROBE=1 python tools/bench_psd_ts.py 200 5000 1 2>&1 | grep probe | cut -c1-260 > gpurun_out/ts4_probe_db.txt
rm -rf /tmp/p4; rocprofv3 --kernel-trace --stats -d /tmp/p4 -o b -- python tools/bench_psd_ts.py 200 5000 3 > /dev/null 2>&1; python tools/rocpd_stats.py $(find /tmp/p4 -name "*.db" | head -1) 2>/dev/null | grep -i "psd_ts" | cut -c1-150 >> gpurun_out/ts4_probe_db.txt
cat gpurun_out/ts4_probe_db.txt
