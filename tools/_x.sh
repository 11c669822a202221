export TMPDIR=/tmp
python -m pytest tests/test_hip_cones.py tests/test_hip_switches.py tests/test_hip_trajectory.py -m gpu -q -x -k "wsos or Wsos or polymin or switch or par" 2>&1 | tail -3
for i in 1 2; do
for c in 5p 5d; do echo "$c lanes: $(python bench.py --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])")"; 
echo "$c PAR=0 : $(HYP_WSOS_PAR=0 python bench.py --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])")"; done; done
