export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_distributed.py -q -x -m gpu -k "overlapped or one_rank" 2>&1 | tail -8
# cost side of the overlap: config 4 through the multi-GPU path with one rank, overlap off / 4 groups
for f in 0 4 0 4; do HYP_DIST_OVERLAP=$f HYP_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --no-secondary --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HYP_DIST_OVERLAP=$f', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()}, 'schur_allreduce_ms', round(d.get('schur_allreduce_ms',-1),3), 'incl_pack', round(d.get('schur_exchange_ms_incl_pack',-1),3), 'small', round(d.get('small_collectives_ms_per_step',-1),3))"; done 2>&1 | tee gpurun_out/dist_overlap.txt
