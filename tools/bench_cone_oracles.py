"""Per-oracle wall times of one HIP cone at a BASELINE.json size (config 3a: EpiNormSpectral 500 x 500 oracle by oracle;
also PosSemidefTri 200 and WSOSInterpNonnegative U = 4845).  Host staging of the C-ABI calls is included.
    python tools/bench_cone_oracles.py epinormspectral 500 500 | possemideftri 200 | wsos"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hypatia_jl_amd as H

kind = sys.argv[1] if len(sys.argv) > 1 else "epinormspectral"
rng = np.random.default_rng(0)
if kind == "epinormspectral":
    d1, d2 = int(sys.argv[2]), int(sys.argv[3])
    cone = H.EpiNormSpectral(d1, d2)
elif kind == "possemideftri":
    s = int(sys.argv[2]); cone = H.PosSemidefTri(s * (s + 1) // 2)
else:
    from oracle import polyutils as pu
    U, pts, Ps = pu.interpolate_box([-1.0] * 4, [1.0] * 4, 8, rng=rng, sample_factor=2)
    cone = H.WSOSInterpNonnegative(U, Ps)
dim = cone.dimension()
pt = np.zeros(dim)
cone.set_initial_point(pt)
pt = pt + 0.01 * rng.standard_normal(dim) * (1 + np.abs(pt)) * (0.1 if kind != "possemideftri" else 0.02)
lib, ctx = H._lib.lib(), H._lib.ctx()


def timed(label, fn, reps=3):
    fn()
    lib.hyp_ctx_synchronize(ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    lib.hyp_ctx_synchronize(ctx)
    print("  %-44s %9.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)


print("%s dim=%d" % (type(cone).__name__, dim))


def reload():
    cone.load_point(pt); cone.reset_data()


def feas():
    reload(); assert cone.is_feas()


timed("load_point + is_feas", feas)
timed("grad (after is_feas)", lambda: (reload(), cone.is_feas(), cone.get_grad()))
cone.load_dual_point(-np.array(cone.get_grad()))
timed("is_dual_feas", lambda: cone.is_dual_feas())
for k in (1, 64):
    arr = np.asfortranarray(rng.standard_normal((dim, k)))
    prod = np.zeros_like(arr)
    timed("hess_prod on %d column(s)" % k, lambda: cone.hess_prod(prod, arr) if k > 1 else cone.hess_prod(prod[:, 0], arr[:, 0]))
d = rng.standard_normal(dim)
timed("dder3", lambda: cone.dder3(d))
if kind == "possemideftri":
    arr = np.asfortranarray(rng.standard_normal((dim, 512)))
    prod = np.zeros_like(arr)
    timed("sqrt_hess_prod on 512 columns", lambda: cone.sqrt_hess_prod(prod, arr))
    timed("inv_hess_prod on 512 columns", lambda: cone.inv_hess_prod(prod, arr))
if dim <= 6000:
    timed("check_numerics + get_proxsqr (explicit H + chol for generic cones)", lambda: (reload(), cone.is_feas(), cone.is_dual_feas(), cone.check_numerics(), cone.get_proxsqr(1.0, True)), reps=2)
