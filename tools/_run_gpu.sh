python -m pytest tests/test_hip_qrcp.py -q -x -m gpu -s 2>&1 | tail -30 > gpurun_out/r02_qrcp.log; cat gpurun_out/r02_qrcp.log
