import ctypes, sys, threading, time
sys.path.insert(0, ".")
import hypatia_jl_amd as H
L = H._lib; lib = L.lib(); ctx0 = L.ctx()
lib.hyp_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
ctx1 = ctypes.c_void_p()
assert lib.hyp_ctx_create(0, ctypes.byref(ctx1)) == 0
lib.hyp_bench_potrf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
n = 4845
def run(ctx, out, reps):
    ms = ctypes.c_double(0)
    lib.hyp_bench_potrf(ctx, n, reps, ctypes.byref(ms))
    out.append(ms.value)
o = []
run(ctx0, o, 5); run(ctx1, o, 5)
print("alone: %.3f ms, %.3f ms" % (o[0], o[1]))
for reps in (20,):
    a, b = [], []
    t0 = time.perf_counter()
    ta = threading.Thread(target=run, args=(ctx0, a, reps)); tb = threading.Thread(target=run, args=(ctx1, b, reps))
    ta.start(); tb.start(); ta.join(); tb.join()
    wall = time.perf_counter() - t0
    print("two at once: %.3f ms and %.3f ms per factorization (event-timed inside each); wall %.1f ms for %d + %d" % (a[0], b[0], wall * 1e3, reps, reps))
