python -m pytest tests -q -x -m gpu > gpurun_out/r02_pytest_full2.log 2>&1; tail -5 gpurun_out/r02_pytest_full2.log
