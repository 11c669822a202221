export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_cfg2_1gpu.json').read()); print(d['ms_per_step'], d['roofline']['frac'], d['phases_ms_per_step'], d['cpu_baseline']['value'])"
python bench.py --config 4 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_1gpu.json
HYP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_rccl_world1.json
python -c "
import json
for f in ('bench_cfg4_1gpu','bench_cfg4_rccl_world1'):
    d=json.loads(open('gpurun_out/%s.json'%f).read()); print(f, d['ms_per_step'], d['roofline']['frac'], d.get('library_exchanges_per_step'))"
