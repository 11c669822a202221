export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_full3.log 2>&1; tail -5 gpurun_out/r02_pytest_full3.log
python bench.py --steps 60 > gpurun_out/bench_tmp.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_tmp.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["phases_ms_per_step"], d["roofline"]["frac"])
PY
