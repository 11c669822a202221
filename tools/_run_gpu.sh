python -m pytest tests/test_hip_cones.py tests/test_golden.py -q -x -m gpu -k "psd or possemidef or golden" 2>&1 | tail -2
python bench.py --config 4 --steps 15 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms_per_step'].items()})
"
