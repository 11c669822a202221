#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
python tools/potrf_accuracy.py make
for v in "" "HYP_POTRF_DEFER=0" "HYP_POTRF_TINV=0" "HYP_POTRF_MFMA=0"; do echo "== $v"; env $v python tools/potrf_accuracy.py eval; done
