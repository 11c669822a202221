# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x -n 1 2>&1 | tail -15
cd /tmp; rm -rf /tmp/prof2; rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o b -- python $R/bench.py --steps 40 --cpu-iters 0 > $R/gpurun_out/x_bench_prof.json 2>/dev/null; cd $R
DB2=$(find /tmp/prof2 -name "*.db" | head -1)
for b in 3 5 8; do ITER_BACK=$b python tools/rocpd_gaps.py $DB2 11000 100000 > gpurun_out/x_timeline_$b.txt 2>/dev/null; head -16 gpurun_out/x_timeline_$b.txt; done
