export TMPDIR=/tmp
for d in 0 1 2 3 4 7; do
HYP_TS_DBG=$d rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ts$d -o b -- python tools/bench_psd_ts.py 200 2500 3 > /dev/null 2>&1
DB=$(find gpurun_out/prof_ts$d -name "*.db" | head -1); python tools/rocpd_stats.py $DB /tmp/ts$d.csv > /dev/null
echo "dbg=$d"; grep psd_ts /tmp/ts$d.csv | cut -d, -f1-4 | cut -c1-90
rm -rf gpurun_out/prof_ts$d
done
