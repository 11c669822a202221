#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1); rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "psd_ts4_kernel" -d /tmp/pmc_$tag -o b -- python bench.py --steps 2 --warmup 1 --cpu-iters 0 > /dev/null 2>&1
  python tools/rocpd_pmc.py $(find /tmp/pmc_$tag -name "*.db" | head -1) > gpurun_out/pmc_ts4_$tag.txt 2>&1; head -12 gpurun_out/pmc_ts4_$tag.txt
done
