# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_hip_switches.py tests/test_polymin_known_minima.py -q -x -k "resident or paired or constant_column or known_minimum" 2>&1 | tail -3
for v in 1 0 1 0; do HYP_DIR_POLL=$v python bench.py --steps 100 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('poll=$v', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],3), d['phases_ms_per_step'], d.get('ms_per_kkt_solve'))"; done
cd /tmp; rm -rf /tmp/prof2; rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o b -- python $R/bench.py --steps 40 --cpu-iters 0 > /dev/null 2>&1; cd $R
DB2=$(find /tmp/prof2 -name "*.db" | head -1)
for b in 3 5; do ITER_BACK=$b python tools/rocpd_gaps.py $DB2 11000 100000 > gpurun_out/x_timeline_$b.txt 2>/dev/null; head -12 gpurun_out/x_timeline_$b.txt; done
