export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_full6.log 2>&1; tail -4 gpurun_out/r02_pytest_full6.log
rm -f gpurun_out/r02_other_configs.jsonl
for c in 3b 5p 5d; do python bench.py --config $c 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl; done
cut -c1-60 gpurun_out/r02_other_configs.jsonl
