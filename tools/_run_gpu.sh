# Scratch script for one gpurun call (`gpurun --timeout N -- 'bash tools/_run_gpu.sh'`): whatever is measured goes under
# gpurun_out/, summaries worth keeping are copied to profiles/.  The round's standard batch:
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; tail -c 400 gpurun_out/bench_cfg2_1gpu.json
