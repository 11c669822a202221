export TMPDIR=/tmp
HYP_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --steps 10 --warmup 2 --secondary-steps 20 > gpurun_out/b4.json 2> gpurun_out/b4.err; python - <<PY
import json
d=json.loads(open('gpurun_out/b4.json').read().strip().splitlines()[-1])
print(d['metric'][:60], d['value'], d['ms_per_step'])
for k in ('weak_config2_block_per_gpu','headline_config2_kshard'):
    r=d.get(k); print(k, {kk: r.get(kk) for kk in ('value','unit','ms_per_step','scaling','error')} if r else None)
PY
tail -3 gpurun_out/b4.err
