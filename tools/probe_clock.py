"""Do latency-bound single-workgroup kernels (potrf diagonal blocks, trsv steps) run at a low clock when
nothing else keeps the chip busy?  Time hyp_dense_potrf (n = 5000) alone and next to a background syrk
loop running on a second context (own stream)."""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hypatia_jl_amd as H
L = H._lib
lib, ctx = L.lib(), L.ctx()
c_vp = ctypes.c_void_p
n = 5000
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n + 50))
A0 = np.asfortranarray(M @ M.T / n + np.eye(n))
x0 = rng.standard_normal(n)

def run_potrf(reps=3):
    ts = []
    for _ in range(reps):
        A = A0.copy(order="F"); x = x0.copy(); info = ctypes.c_int(0)
        lib.hyp_reset_timers(ctx)
        t = time.perf_counter()
        L.check(lib.hyp_dense_posv(ctx, n, A.ctypes.data_as(c_vp), n, x.ctypes.data_as(c_vp), ctypes.byref(info)), "posv")
        ts.append(time.perf_counter() - t)
    return min(ts)

print("posv n=5000 alone (incl. 2x200MB PCIe): %.1f ms" % (run_potrf() * 1e3))
ctx2 = c_vp()
L.check(lib.hyp_ctx_create(0, ctypes.byref(ctx2)), "ctx2")
stop = False
def bg():
    ms = ctypes.c_double(0)
    while not stop:
        lib.hyp_bench_syrk(ctx2, 3000, 4000, 20, ctypes.byref(ms))
th = threading.Thread(target=bg); th.start()
time.sleep(1.0)
print("posv n=5000 with background syrk stream: %.1f ms" % (run_potrf() * 1e3))
stop = True; th.join()
