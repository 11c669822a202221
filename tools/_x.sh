export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_distributed.py -q -x 2>&1 | tail -3
for sh in 8 1; do HYP_BENCH_RANK_SHARE=$sh HYP_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2956$sh bench.py --gpus 1 --config 4 --no-secondary --steps 20 2>/dev/null | tail -1 > gpurun_out/x_dist_share$sh.json; python -c "
import json; d=json.load(open('gpurun_out/x_dist_share$sh.json')); print('sharded driver, world 1, rank share $sh:', round(d['ms_per_step'],2), d['phases_ms_per_step'], d['collectives_per_step'])"; done
HYP_BENCH_RANK_SHARE=8 timeout 600 python bench.py --config 4 --steps 30 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('single process, rank share 8:', round(d['ms_per_step'],2), d['phases_ms_per_step'])"
