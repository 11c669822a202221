export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_full5.log 2>&1; tail -4 gpurun_out/r02_pytest_full5.log
python bench.py > gpurun_out/r02_bench_cfg2_1gpu.json 2> gpurun_out/r02_bench_cfg2_1gpu.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench_cfg2_1gpu.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["phases_ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
