export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_distributed.py -q -x 2>&1 | tail -6
for v in 1 0; do HYP_DIST_RESIDENT=$v HYP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$v bench.py --gpus 1 --config 4 --no-secondary --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dist resident=$v', round(d['ms_per_step'],2), d['phases_ms_per_step'], d['collectives_per_step'])"; done
