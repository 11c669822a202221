export TMPDIR=/tmp
python tools/diag_const3.py 2>&1 | tail -25
