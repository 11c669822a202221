#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg2_1gpu.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['phases_ms_per_step'])"
HYP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 2w --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_cfg2w_rccl_world1.json
HYP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --config 4 --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_rccl_world1.json
python -c "
import json
for f in ('bench_cfg2w_rccl_world1','bench_cfg4_rccl_world1'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['ms_per_step'], d['phases_ms_per_step']['search'], d.get('collectives_per_step'))"
