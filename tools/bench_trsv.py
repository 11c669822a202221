"""Super-block triangular solves in isolation: python tools/bench_trsv.py [n ...] -> HIP-event ms of the plan build, the
one-vector U'U solve and the two-vector one (hyp_bench_trsv); with --dump FILE the solutions are saved for A/B comparisons."""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
import hypatia_jl_amd as H
L = H._lib; lib, ctx = L.lib(), L.ctx()
args = sys.argv[1:]
dump = None
if "--dump" in args:
    k = args.index("--dump"); dump = args[k + 1]; del args[k:k + 2]
out = {}
for n in [int(a) for a in args] or [5000, 4845, 2250]:
    ms = (ctypes.c_double * 4)()
    x = np.zeros(6 * n)
    rc = lib.hyp_bench_trsv(ctx, n, 10, ms, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    print("trsv n=%d: plan build %.3f ms, one vector %.3f ms, two vectors %.3f ms, three vectors %.3f ms (rc %d)" % (n, ms[0], ms[1], ms[2], ms[3], rc))
    out["x%d" % n] = x
if dump:
    np.savez(dump, **out)
