python -m pytest tests/test_hip_cones.py tests/test_hip_solver.py tests/test_golden.py tests/test_hip_baseline_configs.py -q -x -m gpu -k "epinormspectral or generic or ens" 2>&1 | tail -3
python tools/run_config.py --config 3b 2>&1 | tail -1 | cut -c1-330
