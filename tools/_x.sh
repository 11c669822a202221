export TMPDIR=/tmp
python -m pytest tests/test_hip_dense.py tests/test_hip_fullsize.py -m gpu -q -x 2>&1 | tail -2
for v in 1 0 1 0; do
HYP_SYRK_EDGE=$v python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('edge $v', d['ms_per_step'], d['phases_ms_per_step']['syrk'], d['phases_ms_per_step']['update_lhs'], d['roofline']['frac'])"
done
