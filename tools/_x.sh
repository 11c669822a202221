export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; python -c "
import json
d=json.loads(open('gpurun_out/bench_cfg2_1gpu.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['phases_ms_per_step'], d['ms_per_kkt_solve'])"
cd /tmp; rm -rf /tmp/prof2; rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o b -- python $R/bench.py --steps 40 > $R/gpurun_out/bench_under_profiler.json 2>/dev/null; cd $R
DB2=$(find /tmp/prof2 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB2 2>/dev/null | head -40 > gpurun_out/cfg2_kernel_stats.csv
ITER_BACK=3 python tools/rocpd_gaps.py $DB2 0 100000 > gpurun_out/iteration_timeline.txt 2>/dev/null; head -12 gpurun_out/iteration_timeline.txt
python tools/rocpd_timeline.py $DB2 "splitk_reduce_kernel" 140 > gpurun_out/cholesky_timeline.txt 2>/dev/null
for c in 5p 5d 3b; do python bench.py --config $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms_per_step'].items()})"; done
