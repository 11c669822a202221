"""Time the device Bunch-Kaufman (rook) factorization + solve against the Cholesky path and LAPACK dsytrf_rook."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, ".")
import hypatia_jl_amd as H
from oracle import linalg as la
L = H._lib; lib, ctx = L.lib(), L.ctx()
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
kind = "indef"
args = [a for a in sys.argv[1:]]
if args and not args[0].isdigit():
    kind = args.pop(0)     # "indef": random symmetric (rook search + interchanges at nearly every step);
                           # "nearpd": positive definite up to one eigenvalue of -1e-10 (what a failed Cholesky of a Hessian looks like)
for n in [int(a) for a in args] or [1000, 2500, 5000]:
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n)); b = rng.standard_normal(n)
    if kind == "nearpd":
        A = M @ M.T / n
        v = rng.standard_normal(n); v /= np.linalg.norm(v)
        A = np.asfortranarray(A - (v @ A @ v + 1e-10) * np.outer(v, v))
        A = np.asfortranarray(0.5 * (A + A.T))
    else:
        A = np.asfortranarray(M + M.T)
    for rep in range(2):
        Ad, x, info = A.copy(order="F"), b.copy(), ctypes.c_int(0)
        t0 = time.perf_counter()
        L.check(lib.hyp_dense_sysv_rook(ctx, n, fp(Ad), n, fp(x), 1, n, ctypes.byref(info), None, None, None, None), "sysv")
        t1 = time.perf_counter()
    t2 = time.perf_counter(); f = la.bk_rook(A); t3 = time.perf_counter()
    berr = np.linalg.norm(A @ x - b) / (np.linalg.norm(A, 2) * np.linalg.norm(x) + np.linalg.norm(b))
    print(kind + " n=%d device sysv_rook (incl. %.0f MB h2d+d2h) %.1f ms; LAPACK dsytrf_rook %.1f ms; berr %.2e" % (n, 16e-6 * n * n, 1e3 * (t1 - t0), 1e3 * (t3 - t2), berr), flush=True)
