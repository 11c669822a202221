python -m pytest tests/test_hip_dense.py tests/test_hip_cones.py -q -x -m gpu 2>&1 | tail -3
for v in 1 0; do
HYP_POTRF_MFMA=$v python bench.py --steps 30 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MFMA=$v', 'ms/step', round(d['ms_per_step'],3), d['phases_ms_per_step'])
"
done
