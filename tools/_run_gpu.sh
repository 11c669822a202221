export TMPDIR=/tmp
RX='gemm_f64_kernel<true, 4, 1>|splitk_reduce|psd_ts_kernel'
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "$RX" -d gpurun_out/pmc_r02_$tag -o p -- python bench.py --steps 2 --warmup 1 --cpu-iters 0 > gpurun_out/pmc_r02_$tag.log 2>&1
  echo "pass $tag rc=$?"
  DB=$(find gpurun_out/pmc_r02_$tag -name "*.db" | head -1)
  python tools/rocpd_pmc.py $DB > gpurun_out/r02_pmc_$tag.txt 2>&1
  head -30 gpurun_out/r02_pmc_$tag.txt
  rm -rf gpurun_out/pmc_r02_$tag
done
