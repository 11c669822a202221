"""The four passes over the resident G of a KKT solve, in isolation: python tools/bench_gemv.py [n] [side] [HYP_GEMVT_CB values ...]
(the variants are run in subprocesses: the switch is read once per process)"""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
side = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = sys.argv[3:]
if variants and "HYP_BENCH_GEMV_CHILD" not in os.environ:
    np.save("/tmp/_bench_gemv_dummy.npy", np.zeros(1))
    for v in variants:
        env = dict(os.environ, HYP_GEMVT_CB=v, HYP_BENCH_GEMV_CHILD="1")
        print("HYP_GEMVT_CB=%s" % v, flush=True)
        subprocess.run([sys.executable, __file__, str(n), str(side)], env=env)
    sys.exit(0)
import hypatia_jl_amd as H
from hypatia_jl_amd import _lib as L
lib = L.lib()
# a bare system solver over a random G: no model preprocessing needed for the passes
rng = np.random.default_rng(1)
q = side * (side + 1) // 2
cone = H.PosSemidefTri(q)
handles = (ctypes.c_void_p * 1)(cone._h)
h = ctypes.c_void_p()
L.check(lib.hyp_sys_create(L.ctx(), n, 0, q, handles, 1, ctypes.byref(h)), "create")
G = np.asfortranarray(rng.standard_normal((q, n)))
L.check(lib.hyp_sys_load(h, G.ctypes.data_as(ctypes.c_void_p), None, None, None, None), "load")
out = np.zeros(4)
L.check(lib.hyp_sys_bench_gemv(h, 20, L.vec_ptr(out)), "bench_gemv")
for name, ms in zip(("G'X (2 cols)", "G X (2 cols)", "G'x", "G x"), out):
    print("  %-14s %.1f us  %.2f TB/s" % (name, 1e3 * ms, q * n * 8 / ms / 1e9))
