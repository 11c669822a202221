export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_cfg2_1gpu.json').read()); print(d['ms_per_step'], d['roofline']['frac'], d['phases_ms_per_step'], d['cpu_baseline']['value'])"
