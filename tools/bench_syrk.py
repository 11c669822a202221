"""Schur syrk in isolation: python tools/bench_syrk.py [n] [q] [reps]  -> ms and algorithmic TFLOP/s (n^2 q)."""
import ctypes
import sys
sys.path.insert(0, ".")
import hypatia_jl_amd as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 20100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib, ctx = H._lib.lib(), H._lib.ctx()
ms = ctypes.c_double(0)
lib.hyp_bench_syrk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for _ in range(2):
    rc = lib.hyp_bench_syrk(ctx, n, q, reps, ctypes.byref(ms))
print("syrk n=%d q=%d: %.3f ms  %.2f TFLOP/s (rc %d)" % (n, q, ms.value, float(n) * n * q / ms.value / 1e9, rc))
