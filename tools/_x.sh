export TMPDIR=/tmp
for cb in 8 4 8 4; do
HYP_SYRK_EDGE_CB=$cb python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('cb $cb', d['ms_per_step'], d['phases_ms_per_step']['syrk'])"
done
