export TMPDIR=/tmp
for ov in 0 1; do
HYP_EXP_OVERLAP=$ov python bench.py --steps 20 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('overlap $ov', d['ms_per_step'], d['phases_ms_per_step'])"
done
