export TMPDIR=/tmp
for c in 5p 5d; do
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$c -o b -- python tools/run_config.py --config $c > gpurun_out/r02_cfg${c}_run.txt 2>&1
DB=$(find gpurun_out/prof_$c -name "*.db" | head -1); python tools/rocpd_stats.py $DB gpurun_out/r02_cfg${c}_kernel_stats.csv > /dev/null
rm -rf gpurun_out/prof_$c
done
HYP_PROFILE=1 python tools/run_config.py --config 5p > gpurun_out/r02_cfg5p_profile.txt 2>&1
head -30 gpurun_out/r02_cfg5p_kernel_stats.csv | cut -c1-160
tail -40 gpurun_out/r02_cfg5p_profile.txt
