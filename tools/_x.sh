export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_switches.py -q -x -k "accept_time" 2>&1 | tail -2
for v in 1 0 1 0 1 0; do HYP_RP_PREFETCH=$v python bench.py --steps 300 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('prefetch=$v', round(d['ms_per_step'],3), d['phases_ms_per_step']['get_directions'], d['phases_ms_per_step']['search'])"; done
