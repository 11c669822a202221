export TMPDIR=/tmp
for v in "0 0 0" "1 2 0" "1 3 0" "1 4 0" "1 6 0" "1 3 196" "1 2 464" "1 4 100"; do set -- $v
echo "tail=$1 q=$2 extra=$3: $(HYP_SYRK_TAIL=$1 HYP_SYRK_TAIL_Q=$2 HYP_SYRK_TAIL_X=$3 python tools/bench_syrk.py 5000 20100 8 2>&1 | tail -1)"
done
