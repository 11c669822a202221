# The round's measurement batch for one gpurun call (`gpurun --timeout N -- 'bash tools/_run_gpu.sh'`; the test suite: tools/_run_gpu_tests.sh):
# whatever is measured goes under gpurun_out/, summaries worth keeping are copied to profiles/ (named per round) by hand afterwards.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; tail -c 300 gpurun_out/bench_cfg2_1gpu.json
timeout 600 python bench.py --steps 220 --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/bench_cfg2_220steps.json
rm -f gpurun_out/other_configs.jsonl
for c in 3b 5p 5d; do timeout 600 python bench.py --config $c 2>/dev/null | tail -1 >> gpurun_out/other_configs.jsonl; done
timeout 600 python bench.py --config 4 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_1gpu.json
HYP_BENCH_RANK_SHARE=8 timeout 600 python bench.py --config 4 --steps 30 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_rank_share8.json
HYP_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_rccl_world1.json
python -c "
import json
for l in open('gpurun_out/other_configs.jsonl'): d=json.loads(l); print(d['config']['workload'][:50], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('executed_frac'), d.get('solve_plans'))
for f in ('bench_cfg2_220steps','bench_cfg4_1gpu','bench_cfg4_rank_share8','bench_cfg4_rccl_world1'):
    d=json.loads(open('gpurun_out/%s.json'%f).read()); print(f, d['ms_per_step'], d['roofline']['frac'], d['phases_ms_per_step'])"
cd /tmp; rm -rf /tmp/prof2; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o b -- python $R/bench.py --steps 40 > $R/gpurun_out/bench_under_profiler.json 2>/dev/null; cd $R
DB2=$(find /tmp/prof2 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB2 2>/dev/null | head -40 > gpurun_out/cfg2_kernel_stats.csv; head -4 gpurun_out/cfg2_kernel_stats.csv
ITER_BACK=3 python tools/rocpd_gaps.py $DB2 0 100000 > gpurun_out/iteration_timeline.txt 2>/dev/null; head -3 gpurun_out/iteration_timeline.txt
python tools/rocpd_timeline.py $DB2 "splitk_reduce_kernel" 140 > gpurun_out/cholesky_timeline.txt 2>/dev/null
for c in 3b 5p 5d 4; do rm -rf /tmp/prof_$c; cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o b -- python $R/bench.py --config $c $( [ $c = 4 ] && echo "--steps 6 --warmup 2" ) > /dev/null 2>&1; cd $R; python tools/rocpd_stats.py $(find /tmp/prof_$c -name "*.db" | head -1) 2>/dev/null | head -25 > gpurun_out/cfg${c}_kernel_stats.csv; done
timeout 300 python tools/bench_potrf.py > gpurun_out/bench_potrf.txt 2>&1; cat gpurun_out/bench_potrf.txt
timeout 300 python tools/bench_trsv.py 5000 4845 2250 999 > gpurun_out/bench_trsv.txt 2>&1; cat gpurun_out/bench_trsv.txt
# PMC passes: one counter group per run, kernel trace only (no other trace domains)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1); [ $tag = SQ_VALU_MFMA_BUSY_CYCLES ] && tag=SQ_GRBM; rm -rf /tmp/pmc_$tag
  cd /tmp; timeout 900 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "gemm_f64_kernel<true, 4, 1>|psd_ts4_kernel|trsv_onelaunch_kernel|splitk_reduce" -d /tmp/pmc_$tag -o b -- python $R/bench.py --steps 2 --warmup 1 --cpu-iters 0 > /dev/null 2>&1; cd $R
  python tools/rocpd_pmc.py $(find /tmp/pmc_$tag -name "*.db" | head -1) > gpurun_out/pmc_$tag.txt 2>&1; head -12 gpurun_out/pmc_$tag.txt
done
python tools/pmc_summarize.py gpurun_out/pmc
# LDS bank conflicts of the GEMM instances (config 5 primal: the 64 x 64-tile instance; config 2: the Schur syrk)
for cfg in 5p 2; do
  rm -rf /tmp/pmc_lds_$cfg; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "gemm_f64_kernel|trsm_diag_refined_fwd_batched" -d /tmp/pmc_lds_$cfg -o b -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --cpu-iters 0 > /dev/null 2>&1; cd $R
  python tools/rocpd_pmc.py $(find /tmp/pmc_lds_$cfg -name "*.db" | head -1) > gpurun_out/pmc_lds_cfg$cfg.txt 2>&1; head -8 gpurun_out/pmc_lds_cfg$cfg.txt
done
timeout 1500 python tools/parity_margins.py > gpurun_out/parity_margins.txt 2>&1; tail -50 gpurun_out/parity_margins.txt
