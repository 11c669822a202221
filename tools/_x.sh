export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { tag=$1; shift; cd /tmp; rm -rf /tmp/pp_$tag; env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/pp_$tag -o b -- python $R/bench.py --steps 60 --warmup 10 --cpu-iters 0 > /tmp/pp_$tag.json 2>/dev/null; cd $R; tail -1 /tmp/pp_$tag.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('$tag', round(d['ms_per_step'],3), 'dirs %.2f' % p['get_directions'])"; python tools/rocpd_stats.py $(find /tmp/pp_$tag -name "*.db" | head -1) 2>/dev/null | head -70 > gpurun_out/stats_$tag.csv; }
prof a HYP_SYRK_S_PLAIN=1
prof b HYP_SYRK_S_PLAIN=1 HYP_SPLITK_WS_MIN_MB=2048
prof a2 HYP_SYRK_S_PLAIN=1
prof b2 HYP_SYRK_S_PLAIN=1 HYP_SPLITK_WS_MIN_MB=2048
