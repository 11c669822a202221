"""Time PosSemidefTri.sqrt_hess_prod on a (q x ncols) block (the psd_ts kernels): python tools/bench_psd_ts.py [side] [ncols] [reps]"""
import ctypes, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hypatia_jl_amd as H

side = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dim = side * (side + 1) // 2
cone = H.PosSemidefTri(dim)
pt = np.zeros(dim)
cone.set_initial_point(pt)
cone.load_point(pt); cone.load_dual_point(pt); cone.reset_data()
assert cone.is_feas()
cone.get_grad()
rng = np.random.default_rng(0)
arr = np.asfortranarray(rng.standard_normal((dim, ncols)))
prod = np.zeros_like(arr, order="F")
lib, ctx = H._lib.lib(), H._lib.ctx()
# device-resident timing through the system solver would avoid the staging; here the C-ABI call includes
# H2D / D2H of the block, so report the kernel times from rocprof instead when precise numbers are needed
for _ in range(2):
    cone.sqrt_hess_prod(prod, arr)
t0 = time.perf_counter()
for _ in range(reps):
    cone.sqrt_hess_prod(prod, arr)
dt = (time.perf_counter() - t0) / reps
print("sqrt_hess_prod side=%d ncols=%d: %.2f ms per call (incl. host staging)" % (side, ncols, dt * 1e3))
