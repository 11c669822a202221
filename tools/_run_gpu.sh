export TMPDIR=/tmp
for w in 1 2 4; do echo "panel waves $w: $(HYP_PANEL_WAVES=$w python tools/bench_potrf.py 5000 2>&1 | tail -1)"; done
for w in 2 4; do echo "panel waves $w own_cu 0: $(HYP_POTRF_OWN_CU=0 HYP_PANEL_WAVES=$w python tools/bench_potrf.py 5000 2>&1 | tail -1)"; done
