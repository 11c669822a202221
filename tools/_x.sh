# scratch batch (rewritten per call)
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -n 4 > gpurun_out/x_pytest_full.log 2>&1; tail -6 gpurun_out/x_pytest_full.log
