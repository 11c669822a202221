export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" HYP_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config 4 --no-secondary --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_step']; print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in p.items()})"; }
run HYP_BENCH_RANK_SHARE=8 HYP_DIST_OVERLAP=0
run HYP_BENCH_RANK_SHARE=8 HYP_DIST_OVERLAP=4
run HYP_BENCH_RANK_SHARE=8 HYP_DIST_OVERLAP=0
run HYP_BENCH_RANK_SHARE=8 HYP_DIST_OVERLAP=4
run HYP_BENCH_RANK_SHARE=8 HYP_DIST_OVERLAP=3
cd /tmp; rm -rf /tmp/p1; HYP_DIST_OVERLAP=4 HYP_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 $R/bench.py --gpus 1 --config 4 --no-secondary --steps 3 --warmup 2 > /dev/null 2>&1; cd $R
for DB in $(find /tmp/p1 -name "*.db"); do python tools/rocpd_timeline.py $DB "tri_pack_rows_kernel" 5 18 2>/dev/null | grep -v "axpby\|dot_\|lincomb\|psd_ts5" ; echo ----; done > gpurun_out/overlap_timeline.txt 2>&1
