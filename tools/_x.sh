export TMPDIR=/tmp
for f in 2 3; do echo "HYP_TRSV_ADAPT=$f"; python tools/diag_traj.py cfg5dw_1 "{\"HYP_TRSV_ADAPT\":\"$f\"}" 2>&1 | sed -n 10,13p | cut -c1-250; done
for f in 2 3; do echo "reference route HYP_TRSV_ADAPT=$f"; python tools/diag_traj.py cfg5dw_1 "{\"HYP_TRSV_ADAPT\":\"$f\",\"HYP_ENS_CLOSED_INV\":\"0\",\"HYP_PROX_LB\":\"0\",\"HYP_ENS_PREFETCH\":\"0\",\"HYP_WSOS_PAR\":\"0\",\"HYP_BK_HYBRID\":\"0\"}" 2>&1 | sed -n 10,13p | cut -c1-250; done
