# GPU test suite of a round (its own gpurun call: a hung test must not eat the measurement batch's time)
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
