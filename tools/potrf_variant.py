"""One blocked Cholesky under the current HYP_POTRF_* switches: prints sha256 of the factor's upper triangle and its backward error
(|| U'U - A || / (n || A ||)) for a seeded matrix: python tools/potrf_variant.py n [cond] -- used by tests/test_hip_dense.py to compare
the forms of the diagonal-block kernel (bitwise where the arithmetic is the same, to LAPACK's backward error where it is not)."""
import ctypes, hashlib, json, sys
import numpy as np
sys.path.insert(0, ".")
import hypatia_jl_amd as H
L = H._lib; lib, ctx = L.lib(), L.ctx()
n = int(sys.argv[1]); cond = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rng = np.random.default_rng(n)
if cond > 0:
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * np.logspace(0, -np.log10(cond), n)) @ Q.T
    A = np.asfortranarray(0.5 * (A + A.T))
else:
    M = rng.standard_normal((n, n + 5))
    A = np.asfortranarray(M @ M.T + 0.5 * np.eye(n))
Ad = A.copy(order="F")
info = ctypes.c_int(-1)
L.check(lib.hyp_dense_potrf(ctx, n, Ad.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(info)), "potrf")
U = np.triu(Ad)
print(json.dumps({"info": info.value, "sha": hashlib.sha256(np.ascontiguousarray(U).tobytes()).hexdigest(),
                  "berr": float(np.linalg.norm(U.T @ U - A) / (n * np.linalg.norm(A)))}))
