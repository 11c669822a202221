export TMPDIR=/tmp
for v in 8 0 8 0 8 0; do
HYP_TS_FEW=$v python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('few $v', d['ms_per_step'], d['phases_ms_per_step']['update_lhs'], d['phases_ms_per_step']['get_directions'], d['phases_ms_per_step']['search'])"
done
