export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" HYP_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config 4 --no-secondary --steps 12 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_step']; print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in p.items()})"; }
run HYP_DIST_OVERLAP=0
run HYP_DIST_OVERLAP=4
run HYP_DIST_OVERLAP=8
run HYP_DIST_OVERLAP=4 HYP_DIST_OVERLAP_OLD=1
timeout 900 python -m pytest tests/test_hip_distributed.py -q -m gpu -x 2>&1 | tail -3
