"""config 4 (n = 5000, 64 PSD cones of side 80): per-iteration wall time and the G products of the convergence check"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import instances as I
import hypatia_jl_amd as H
inst = I.psd_blocks(5000, [80] * 64, seed=1, dtype=np.float32)
s = H.Solver(verbose=False, init_use_indirect=True)
s.load(H.make_model(inst)); s.setup()
sv = s.syssolver
for _ in range(3): s.iterate()
t0 = time.perf_counter()
for _ in range(6): s.iterate()
print("config 4: %.1f ms per iteration" % (1e3 * (time.perf_counter() - t0) / 6), flush=True)
def t(f):
    t0 = time.perf_counter(); f(); return 1e3 * (time.perf_counter() - t0)
for rep in range(2):
    s.iterate(); a = t(lambda: sv.mul_G(True, s.point.z)); a2 = t(lambda: sv.mul_G(True, s.point.z)); a3 = t(lambda: sv.mul_G(False, s.point.x))
    print("first G'z after an iteration %.2f ms, again %.2f ms, G x %.2f ms" % (a, a2, a3), flush=True)
