export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = old ]; then export HYP_LIB_PATH=$R/tools/_bin/libhyp_ts4read2.so; else unset HYP_LIB_PATH; fi
  echo $v; python tools/bench_potrf.py 2>&1 | tail -3
done
unset HYP_LIB_PATH
python tools/potrf_accuracy.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_dense.py -q -x -k "potrf or chol" 2>&1 | tail -2
