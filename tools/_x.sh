export TMPDIR=/tmp
python tools/diag_traj.py cfg5dw_1 2>&1 | tail -30
python tools/diag_traj.py cfg5dw_1 '{"HYP_TRSV_ONE_LAUNCH":"0"}' 2>&1 | tail -14
python - <<'P'
import sys
sys.path.insert(0,'tests')
import trajectory_harness as T
print(T.REFERENCE_ROUTE)
P
python tools/diag_traj.py cfg5pw_1 2>&1 | tail -30
