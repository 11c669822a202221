export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_k20.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_k20.json').read()); print(d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['config']['algorithm'])"
HYP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --no-secondary --steps 20 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_rccl_world1.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_cfg4_rccl_world1.json').read()); print('rccl world1', d['ms_per_step'], d.get('collectives_per_step'), d.get('library_exchanges_per_step'))"
