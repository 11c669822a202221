# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_hip_switches.py tests/test_hip_dense.py -q -x -k "syrk" 2>&1 | tail -3
for v in 1 0 1 0; do HYP_SYRK_FUSED_REDUCE=$v python bench.py --steps 100 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fused reduce=$v', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],3), d['phases_ms_per_step'])"; done
