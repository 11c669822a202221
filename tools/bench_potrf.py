"""Blocked Cholesky in isolation: python tools/bench_potrf.py [n ...] -> wall ms of hyp_dense_potrf minus the transfers is not
separable, so the device time is read from HIP events inside hyp_bench_potrf."""
import ctypes, sys
sys.path.insert(0, ".")
import hypatia_jl_amd as H
L = H._lib; lib, ctx = L.lib(), L.ctx()
lib.hyp_bench_potrf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for n in [int(a) for a in sys.argv[1:]] or [5000, 4845, 2250]:
    ms = ctypes.c_double(0)
    rc = lib.hyp_bench_potrf(ctx, n, 10, ctypes.byref(ms))
    print("potrf n=%d: %.3f ms  %.2f TFLOP/s (rc %d)" % (n, ms.value, n ** 3 / 3 / ms.value / 1e9, rc))
