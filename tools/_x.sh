export TMPDIR=/tmp
for sb in 1024 2048 1024 2048; do
HYP_TRSV_SB=$sb python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('sb $sb', d['ms_per_step'], d['phases_ms_per_step']['update_lhs'], d['phases_ms_per_step']['get_directions'], d['phases_ms_per_step']['search'])"
done
HYP_TRSV_SB=2048 python -m pytest tests/test_hip_dense.py tests/test_hip_fullsize.py -m gpu -q -x 2>&1 | tail -2
