# scratch batch (rewritten per call): determinism of the fenced scalar mirror
export TMPDIR=/tmp
python tools/stress_determinism.py 250 rosenbrock 2>&1 | tail -1
python tools/stress_determinism.py 60 mixed 2>&1 | tail -1
HYP_DIR_POLL=0 python tools/stress_determinism.py 100 rosenbrock 2>&1 | tail -1
timeout 900 python -m pytest tests/test_hip_switches.py -q -x -k "resident or paired" 2>&1 | tail -2
for v in 1 0; do HYP_DIR_POLL=$v python bench.py --steps 100 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('poll=$v', round(d['ms_per_step'],3), d['phases_ms_per_step']['get_directions'])"; done
