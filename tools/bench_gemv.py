"""The four passes over the resident G of a KKT solve, in isolation: python tools/bench_gemv.py [n] [side]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
import bench
import hypatia_jl_amd as H
from hypatia_jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
side = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s = H.Solver(verbose=False)
s.load(H.make_model(bench.gen_instance(n, [side], 1)))
s.setup()
out = np.zeros(4)
L.check(L.lib().hyp_sys_bench_gemv(s.syssolver._h, 20, L.vec_ptr(out)), "bench_gemv")
q = s.model.q
for name, ms in zip(("G'X (2 cols)", "G X (2 cols)", "G'x", "G x"), out):
    print("%-14s %.1f us  %.2f TB/s" % (name, 1e3 * ms, q * n * 8 / ms / 1e9))
