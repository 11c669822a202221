export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_solver.py tests/test_hip_trajectory.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2 3; do
python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print(d['ms_per_step'], d['phases_ms_per_step']['update_lhs'], d['phases_ms_per_step']['get_directions'], d['phases_ms_per_step']['search'])"
done
