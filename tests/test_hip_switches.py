"""The line search evaluates independent pieces of a candidate side by side on the two streams; every such
path has a switch that restores the one-after-the-other form.  The switches are read once per process, so each setting runs in
a process of its own; the solves must agree: same status, same iteration count, objective to the solver's own tolerance."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import json, sys
sys.path.insert(0, %r)
import hypatia_jl_amd as H
from oracle import instances as I
name = sys.argv[1]
if name == "matrixcompletion":
    inst = I.matrixcompletion(12, 20, seed=3)
elif name == "psd_single":
    inst = I.psd_blocks(40, [48], seed=5)
elif name == "psd_single_wide":             # one cone of side 72 (three 16-column tile rows and a ragged one)
    inst = I.psd_blocks(90, [72], seed=9)
elif name == "psd_run":                     # five equal cones: one run is the whole model (the grouped path of config 4)
    inst = I.psd_blocks(60, [24, 24, 24, 24, 24], seed=11)
elif name == "psd_trio":                    # three equal cones: too few for a grouped run, the per-cone sweep of check_cone_points
    inst = I.psd_blocks(50, [20, 20, 20], seed=12)
elif name == "psd_many_small":              # 300 cones of side 3: more (candidate, cone) pairs than the screen's pinned blocks hold at 18 candidates
    inst = I.psd_blocks(60, [3] * 300, seed=13)
elif name == "psd_pair":
    inst = I.psd_blocks(50, [40, 33], seed=6)
elif name == "psd_plan":                     # n = 600: the factor has a super-block solve plan (n >= 512)
    inst = I.psd_blocks(600, [36, 20], seed=7)
elif name == "psd_wide_plan":                # one cone of side 63 (q = 2016, a multiple of 4: the one-pass kernel applies) over n = 600: a solve plan, the resident line search, passes over G in one sweep
    inst = I.psd_blocks(600, [63], seed=17)
elif name == "psd_split":                    # n = 1900, q = 4656: a Schur complement large enough to be formed and factored in two column groups
    inst = I.psd_blocks(1900, [96], seed=19)
elif name == "psd_smoke":                    # __graft_entry__.smoke()'s instance
    inst = I.psd_blocks(40, [12, 7], seed=5)
elif name == "polymin_primal":
    inst = I.polymin(2, 3, True, seed=2)
elif name == "polymin_dual":
    inst = I.polymin(2, 3, False, seed=2)
elif name == "polymin_large_primal":
    inst = I.polymin(3, 7, True, seed=3)     # U = binomial(17, 3) = 680
elif name == "polymin_large_dual":
    inst = I.polymin(3, 7, False, seed=3)
else:
    inst = I.KNOWN_ANSWER[name]()
import os
stepper = H.CombinedStepper(use_max_prox=False) if os.environ.get("HYP_TEST_SUM_PROX") == "1" else None   # (search.jl:8-39)
s = H.Solver(default_tol_relax=10, stepper=stepper)
trace = []
s.iter_callback = lambda sv: trace.append((sv.primal_obj, sv.dual_obj, sv.mu, sv.point.tau, sv.x_feas, sv.z_feas))
s.load(H.make_model(inst))
s.solve()
screens = s.syssolver.search_screen_stats() if hasattr(s.syssolver, "search_screen_stats") else (0, 0)
print(json.dumps({"status": s.get_status(), "iters": s.get_num_iters(), "obj": s.get_primal_obj(), "trace": trace,
                  "trials": s.stepper.searcher.n_trials, "screens": screens}))
"""


def _run(name, env_extra, with_stderr=False):
    from conftest import child_env
    env = child_env(env_extra)   # (HYP_PERSISTENT=1 only if this pytest process holds a context -- and with it possibly the device's lock)
    r = subprocess.run([sys.executable, "-c", SNIPPET % ROOT, name], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    if with_stderr:
        out["stderr"] = r.stderr
    return out


@pytest.mark.parametrize("name,switch", [
    ("matrixcompletion", "HYP_ENS_PREFETCH"),            # dual decomposition on the helper stream next to the primal one
    ("epinormspectral3_3x4_dual", "HYP_ENS_PREFETCH"),
    ("polymin_primal", "HYP_WSOS_PAR"),                   # the K feasibility chains on both streams, one read-back
    ("polymin_dual", "HYP_WSOS_PAR"),
    ("wsosinterpnonnegative2", "HYP_WSOS_PAR"),
    ("psd_single", "HYP_PROX_LB"),                        # PosSemidefTri: the proximity value from U Z U' / sqrt(mu) - I, no inverse
    ("psd_pair", "HYP_PROX_LB"),
    ("matrixcompletion", "HYP_PROX_LB"),                  # EpiNormSpectral: rejected before either decomposition is started
    ("epinormspectral3_3x4_dual", "HYP_PROX_LB"),
    ("polymin_large_primal", "HYP_PROX_LB"),              # candidates rejected on a lower bound of the proximity value (U = 680 >= 512)
    ("polymin_large_dual", "HYP_PROX_LB"),
])
def test_side_by_side_candidate_evaluation_matches_sequential(name, switch):
    on = _run(name, {switch: "1"})
    off = _run(name, {switch: "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"]
    assert abs(on["obj"] - off["obj"]) <= 1e-6 * (1 + abs(off["obj"]))


@pytest.mark.parametrize("name", ["matrixcompletion", "epinormspectral3_3x4_dual", "epinormspectral2_primal", "epinormspectral4_dual"])
def test_one_workgroup_spectral_oracles_solve_the_same_problem(name):
    """HYP_ENS_FUSED (round 5, default 1): EpiNormSpectral's feasibility test, gradient + auxiliary matrices, Hessian product and closed-form
    inverse as ONE launch of one workgroup each (d1 <= 64) instead of 7-22 launches.  The Cholesky of Z and the scalar sums are formed
    in another order than the launch chains' (the products and the solves with Z are the same wavefront programs), so late iterates
    differ in the last bits and a borderline step of the line search may fall the other way: same status and optimum, iteration
    counts within one (the oracle-compared trajectories of tests/test_hip_trajectory.py run both forms)."""
    on = _run(name, {"HYP_ENS_FUSED": "1"})
    off = _run(name, {"HYP_ENS_FUSED": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert abs(on["iters"] - off["iters"]) <= 1
    assert abs(on["obj"] - off["obj"]) <= 1e-6 * (1 + abs(off["obj"]))
    k = min(len(on["trace"]), len(off["trace"])) // 2           # the first half of the solve: the same iterates to 1e-9
    for a, b in zip(on["trace"][:k], off["trace"][:k]):
        assert all(abs(x - y) <= 1e-9 * (1 + abs(y)) for x, y in zip(a, b))


@pytest.mark.parametrize("name", ["matrixcompletion", "epinormspectral3_3x4_dual", "epinormspectral2_primal", "epinormspectral4_dual"])
def test_dual_feasibility_decided_on_bounds_changes_no_bit(name):
    """HYP_ENS_DUAL_DECIDE (round 5, default 1): the decomposition behind EpiNormSpectral's dual feasibility test stops sweeping once
    rigorous bounds of the nuclear norm lie on one side of the epigraph variable.  Only the yes / no of the test enters the search:
    same candidates accepted, same iterates to the last bit."""
    on = _run(name, {"HYP_ENS_DUAL_DECIDE": "1"})
    off = _run(name, {"HYP_ENS_DUAL_DECIDE": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["trace"] == off["trace"] and on["trials"] == off["trials"]


@pytest.mark.parametrize("name", ["polymin_primal", "polymin_large_primal", "polymin_large_dual"])
def test_one_lane_per_wsos_chain_changes_no_bit(name):
    """HYP_LANES (round 4, default 3): the K independent chains of a WSOS cone's gradient, of the proximity bound's Gram products and of
    its Hessian-vector product each run on a stream of their own instead of two streams; a chain's kernels and partial results do
    not depend on where it runs."""
    many = _run(name, {"HYP_LANES": "6"})
    two = _run(name, {"HYP_LANES": "2"})
    assert many["status"] == two["status"] == "Optimal"
    assert many["trace"] == two["trace"] and many["trials"] == two["trials"]


@pytest.mark.parametrize("name", ["psd_single", "psd_pair", "matrixcompletion", "polymin_dual"])
def test_constant_column_as_third_column_solves_the_same_problem(name):
    """HYP_CONST_COL3 (on by default since round 3): the constant column of update_lhs rides along with the
    first pair of directions.  Same solve with the switch off: status, iteration count, optimum (bitwise traces: the test at
    the end of this file)"""
    on = _run(name, {"HYP_CONST_COL3": "1"})
    off = _run(name, {"HYP_CONST_COL3": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"]
    assert abs(on["obj"] - off["obj"]) <= 1e-9 * (1 + abs(off["obj"]))


def test_constant_column_in_the_first_paired_solve_changes_no_bit():
    """default since round 3: the constant column of update_lhs (qrchol.jl:191-197) rides along with the first pair of directions as
    a third column -- right-hand side, passes over G, cone products, triangular solves (HYP_CONST_COL3; with that off its
    triangular solves still do: HYP_CONST_TRI3).  Every kernel on the path computes a column with the sums it gets alone, so every
    iterate of a solve must be the one the separate constant solve gives, to the last bit -- on a model with a super-block solve
    plan (n = 600) and on one without (the smoke instance, whose solve took 15 instead of 11 iterations while one kernel differed)"""
    for name in ("psd_plan", "psd_smoke"):
        runs = [_run(name, env) for env in ({}, {"HYP_CONST_COL3": "0"}, {"HYP_CONST_COL3": "0", "HYP_CONST_TRI3": "0"})]
        assert all(r["status"] == "Optimal" for r in runs)
        assert runs[0]["iters"] == runs[1]["iters"] == runs[2]["iters"] >= 8
        assert runs[0]["trace"] == runs[1]["trace"] == runs[2]["trace"], name


@pytest.mark.parametrize("name", ["psd_plan", "polymin_large_primal", "polymin_large_dual"])
def test_one_launch_triangular_sweeps_change_no_bit(name):
    """round 5, HYP_TRSV_ONE_LAUNCH (csrc/trsv_onelaunch.hip): the super-block triangular solves of the Schur factor (potrs of
    qrchol.jl:66-69) and of a generic cone's Hessian factor (Cones.jl:113-118) as ONE persistent launch for both sweeps (default), as
    one launch per sweep (=1) and as the round-2 chain of one launch per product (=0).  A wavefront forms a column's product with the
    chain kernels' loads, accumulators and shuffle tree, so every iterate must be the same to the last bit.  Models with a solve plan:
    n = 600 (Schur factor) and U = 680 (WSOS Hessian factor, one- and two-column solves of the proximity test, both forms)"""
    runs = [_run(name, env) for env in ({}, {"HYP_TRSV_ONE_LAUNCH": "1"}, {"HYP_TRSV_ONE_LAUNCH": "0"})]
    assert all(r["status"] == "Optimal" for r in runs)
    assert runs[0]["iters"] == runs[1]["iters"] == runs[2]["iters"] >= 8
    assert runs[0]["trace"] == runs[2]["trace"] and runs[1]["trace"] == runs[2]["trace"], name
    assert runs[0]["trials"] == runs[1]["trials"] == runs[2]["trials"]


@pytest.mark.parametrize("name", ["psd_single", "psd_single_wide", "psd_run", "psd_trio"])
def test_screened_schedule_walk_changes_no_bit(name):
    """round 3: for a model of one PosSemidefTri cone (or of one run of equal ones: batch = candidates x cones) the schedule walk of search_alpha screens all remaining candidates side by
    side (HYP_SEARCH_SCREEN) -- the tests that reject, batched over the candidates -- and only survivors go
    through the sequential acceptance test; in the fused step the candidates are formed on the device from the directions
    step_directions left there (HYP_SEARCH_RESIDENT; off: formed on the host and uploaded).  The screen may only reject what the
    sequential test rejects: same accepted step in every iteration, hence the same iterates to the last bit, and the same number
    of candidates visited -- in all three forms."""
    on = _run(name, {})
    host = _run(name, {"HYP_SEARCH_RESIDENT": "0"})
    lb = _run(name, {"HYP_SCREEN_SKIP_LB": "0"})   # (a survivor's proximity lower bound evaluated again by the sequential test)
    off = _run(name, {"HYP_SEARCH_SCREEN": "0"})
    assert on["status"] == host["status"] == lb["status"] == off["status"] == "Optimal"
    assert on["iters"] == host["iters"] == lb["iters"] == off["iters"] >= 8
    assert on["trace"] == off["trace"]
    assert host["trace"] == off["trace"]
    assert lb["trace"] == off["trace"]
    assert on["trials"] == host["trials"] == lb["trials"] == off["trials"]
    assert off["screens"] == [0, 0]
    for r in (on, host, lb):
        assert r["screens"][0] >= r["iters"] and r["screens"][1] > 0   # (it ran, and it rejected something)


@pytest.mark.parametrize("name", ["psd_single", "psd_run"])
def test_screened_schedule_walk_with_summed_proximity(name):
    """the same with use_max_prox = false (search.jl:126-131: the cones' proximity values add up instead of the largest one
    counting): the screen aggregates as the sequential test does, the iterates are bitwise those of the sequential walk"""
    on = _run(name, {"HYP_TEST_SUM_PROX": "1"})
    off = _run(name, {"HYP_TEST_SUM_PROX": "1", "HYP_SEARCH_SCREEN": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"] >= 8
    assert on["trace"] == off["trace"]
    assert on["trials"] == off["trials"]
    assert on["screens"][0] >= on["iters"] and on["screens"][1] > 0 and off["screens"] == [0, 0]


def test_screen_batches_shrink_for_models_with_many_cones():
    """round-3 advisor finding: with 226 or more equal PSD cones the 18-candidate batch overran the pinned read-back blocks and
    search_alpha raised in iteration 1 (the sequential walk had accepted such models).  The batch size now follows the model
    (SysSolver::screen_kmax: 13 candidates at 300 cones), and a byte budget too small for two candidates (HYP_SCREEN_MB) turns the
    screen off instead of failing an allocation.  Same iterates, to the bit, in all three forms."""
    on = _run("psd_many_small", {})
    tiny = _run("psd_many_small", {"HYP_SCREEN_MB": "1"})     # 1 MiB: 300 cones x 6 matrices x 9 doubles + candidates -> fewer per batch
    off = _run("psd_many_small", {"HYP_SEARCH_SCREEN": "0"})
    assert on["status"] == tiny["status"] == off["status"] == "Optimal"
    assert on["iters"] == tiny["iters"] == off["iters"] >= 8
    assert on["trace"] == off["trace"] and tiny["trace"] == off["trace"]
    assert on["trials"] == tiny["trials"] == off["trials"]
    assert on["screens"][0] >= on["iters"] and on["screens"][1] > 0 and off["screens"] == [0, 0]


@pytest.mark.parametrize("name", ["polymin_large_primal", "polymin_large_dual"])
def test_wsos_candidate_screen_changes_no_bit(name):
    """round 4 (late): for a model of one large WSOSInterpNonnegative cone the next candidates of the schedule go through the
    feasibility chains, the gradient's triangular solves and the proximity lower bound TOGETHER (WsosCone::screen_batch,
    csrc/wsos_screen.hip; HYP_WSOS_SCREEN = candidates per batch, 0 = off); candidates it reports rejected are stepped over, the
    others are tested by the unchanged sequential code.  The screen may only reject what the sequential test rejects: same
    accepted step in every iteration, the same iterates to the last bit, the same number of candidates visited.
    HYP_WSOS_SCREEN_CHECK=1 evaluates every screened-out candidate sequentially as well and raises on a disagreement."""
    on = _run(name, {})
    wide = _run(name, {"HYP_WSOS_SCREEN": "8"})
    off = _run(name, {"HYP_WSOS_SCREEN": "0"})
    chk = _run(name, {"HYP_WSOS_SCREEN_CHECK": "1"})
    assert on["status"] == wide["status"] == off["status"] == chk["status"] == "Optimal"
    assert on["iters"] == wide["iters"] == off["iters"] == chk["iters"] >= 8
    assert on["trace"] == off["trace"] and wide["trace"] == off["trace"] and chk["trace"] == off["trace"]
    assert on["trials"] == wide["trials"] == off["trials"] == chk["trials"]
    assert off["screens"] == [0, 0]
    for r in (on, wide, chk):
        assert r["screens"][0] > 0 and r["screens"][1] > 0   # (it ran, and it rejected something)


@pytest.mark.parametrize("name", ["psd_plan", "psd_smoke", "psd_pair", "psd_run", "matrixcompletion", "polymin_large_primal", "polymin_large_dual"])
def test_device_resident_direction_scalars_change_no_bit(name):
    """round 6, HYP_DIR_RESIDENT (default on; csrc/directions_multi.hip): the tau / kap of every solve of step_directions are formed
    on the device by a one-thread kernel (the host's operations in the host's order) and read from device memory by the kernels
    behind it; the acceptance tests of the third-order terms (steppers/common.jl:37-55, 96-113) are taken on the device; the
    Cholesky's info word is read together with the first pair's scalars instead of being waited for.  With the switch off the
    same kernels run with host scalars and a host round trip at each of those points: every iterate must agree to the last bit
    -- PSD models with and without a solve plan (paired refinement), a run of equal cones, a spectral cone and a WSOS cone in both
    forms (Bunch-Kaufman fall-backs behind the optimistic Cholesky read)"""
    on = _run(name, {"HYP_DIR_RESIDENT": "1"})
    off = _run(name, {"HYP_DIR_RESIDENT": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"] >= 6
    assert on["trace"] == off["trace"], name
    assert on["trials"] == off["trials"]


@pytest.mark.parametrize("name", ["psd_plan", "psd_pair", "psd_run"])
def test_paired_refinement_solves_the_same_problem(name):
    """round 6, HYP_REFINE_PAIRED (default on): the refinement steps of get_directions (common.jl:38-72) of the two columns of a pair
    through the pair's own column routines -- two columns that both need a step share its passes over G and the factor, a step
    costs one host round trip -- instead of one column after the other through the single-column routines.  The products of a
    refined direction's residual are formed in one pass over G (another summation order than the two one-sided passes), so the
    iterates agree to rounding, not bitwise: same status and iteration count, the objective trace to 1e-9"""
    on = _run(name, {"HYP_REFINE_PAIRED": "1"})
    off = _run(name, {"HYP_REFINE_PAIRED": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert abs(on["iters"] - off["iters"]) <= 1
    assert abs(on["obj"] - off["obj"]) <= 1e-9 * (1 + abs(off["obj"]))
    for a, b in list(zip(on["trace"], off["trace"]))[: min(on["iters"], off["iters"]) // 2]:
        assert abs(a[0] - b[0]) <= 1e-8 * (1 + abs(b[0]))


@pytest.mark.parametrize("name", ["psd_plan", "psd_pair", "psd_single_wide", "psd_smoke"])
def test_paired_third_order_terms_change_no_bit(name):
    """round 6, HYP_DDER3_PAIRED (default on): the right-hand sides of the adjusted directions (steppers/common.jl:37-55, 96-113) of
    a PosSemidefTri cone -- H dir of both columns as ONE two-column product, the five GEMMs of dder3 (possemideftri.jl:197-207)
    stacked / batched over the two columns, the four scalar products in one launch.  Every kernel on that path forms a column with
    the sums it forms alone: every iterate must be the separate launches', to the last bit"""
    on = _run(name, {"HYP_DDER3_PAIRED": "1"})
    off = _run(name, {"HYP_DDER3_PAIRED": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"] >= 8
    assert on["trace"] == off["trace"], name
    assert on["trials"] == off["trials"]


SYRK_SNIPPET = r"""
import ctypes, hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
from hypatia_jl_amd import _lib as L
lib = L.lib()
ctx = ctypes.c_void_p()
assert lib.hyp_ctx_create(0, ctypes.byref(ctx)) == 0
N, K = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(N + K)
A = np.asfortranarray(rng.standard_normal((K, N), dtype=np.float32).astype(np.float64))
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
out = []
for rep in range(3):
    C = np.full((N, N), 3.0, order="F")
    L.check(lib.hyp_dense_syrk(ctx, N, K, fp(A), K, fp(C), N), "syrk")
    out.append(hashlib.sha256(C.tobytes()).hexdigest())
cols = [0, 127, 128, N // 2, N - 1]
ref = A.T @ A[:, cols]
err = max(float(np.max(np.abs(C[:c + 1, c] - ref[:c + 1, t]))) for t, c in enumerate(cols))
print(json.dumps({"hashes": out, "err": err, "scale": float(np.max(np.abs(ref)))}))
"""


@pytest.mark.parametrize("N,K", [(1300, 4100), (2056, 8200), (5000, 20100)])
def test_syrk_reduction_inside_the_product_changes_no_bit(N, K):
    """round 6, HYP_SYRK_FUSED_REDUCE=1 (measured slower than the reduction kernel, default off; csrc/gemm_f64_kernel.hpp): the split-K slices of the Schur syrk (dense.jl:80-86,
    qrchol.jl:234) count their arrivals per tile and the last one adds the partial sums -- in slice order, from device-coherent
    loads -- and writes C, instead of a reduction kernel behind the product.  Same sums in the same order: the matrix must be the
    separate kernel's to the last bit, launch after launch (an arrival counted before its partial sums were visible would show as a
    changing hash), at sizes with and without the cut last round (tail sub-slices) and the thin edge columns"""
    runs = {}
    for v in ("1", "0"):
        from conftest import child_env
        env = child_env({"HYP_SYRK_FUSED_REDUCE": v})
        r = subprocess.run([sys.executable, "-c", SYRK_SNIPPET % ROOT, str(N), str(K)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[v] = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(set(runs["1"]["hashes"])) == 1 and len(set(runs["0"]["hashes"])) == 1
    assert runs["1"]["hashes"][0] == runs["0"]["hashes"][0]
    assert runs["1"]["err"] <= 1e-12 * K * max(1.0, runs["1"]["scale"])


@pytest.mark.parametrize("name", ["psd_single_wide", "psd_wide_plan"])
def test_residual_products_queued_at_accept_time_change_no_bit(name):
    """round 6, HYP_RP_PREFETCH (default on): G' z, G x + s, h' z and z' s of the NEXT iterate (calc_convergence_params, Solvers.jl:425-483)
    are queued the moment the line search accepts a candidate -- x formed on the device with the host mirror's roundings -- and handed
    out by residual_products only if the point it is called with is bit for bit the one they were formed from.  Same kernels on the
    same numbers: every iterate agrees to the last bit with the switch off, and the prefetched products were actually used"""
    runs = {}
    for v in ("1", "0"):
        from conftest import child_env
        env = child_env({"HYP_RP_PREFETCH": v, "HYP_RP_PREFETCH_STATS": "1"})
        r = subprocess.run([sys.executable, "-c", SNIPPET % ROOT, name], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[v] = (json.loads(r.stdout.strip().splitlines()[-1]), r.stderr)
    on, off = runs["1"][0], runs["0"][0]
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"] >= 8
    assert on["trace"] == off["trace"], name
    import re
    m = re.search(r"handed out (\d+), recomputed (\d+)", runs["1"][1])
    assert m and int(m.group(1)) >= on["iters"] - 2 and int(m.group(2)) == 0, runs["1"][1][-500:]


@pytest.mark.parametrize("name", ["psd_single_wide", "psd_run"])
def test_x_rows_only_download_of_the_directions_changes_no_bit(name):
    """round 6, hyp_sys_set_direction_rows (HYP_DIRS_X_ONLY, default on where the line search runs on the resident directions): only the x
    rows and tau / kap of the four directions are copied to the host -- all update_stepper_points (combined.jl:124-170) reads there once
    the accepted candidate's z / tau / s / kap rows come back from the search.  Same iterates to the last bit as with whole vectors"""
    on = _run(name, {"HYP_DIRS_X_ONLY": "1"})
    off = _run(name, {"HYP_DIRS_X_ONLY": "0"})
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"] >= 8
    assert on["trace"] == off["trace"], name
    assert on["trials"] == off["trials"]


def test_schur_complement_formed_and_factored_in_two_column_groups_changes_no_bit():
    """round 6, HYP_CHOL_SPLIT=<percent> (measured slower, default off; dense.hip: schur_split_begin / _finish): the Schur syrk (qrchol.jl:234)
    forms the leading block columns first, their Cholesky factorization (dense.jl:194-215) runs on other queues while the product
    forms the rest, and the block steps of the leading part are replayed for the late columns behind it.  Every entry of the factor
    sees the operations of the one-piece factorization in the same order; with the same K slices on both sides (the one-piece
    product's cut last round off, its slice count forced) the Schur matrix is the same too: the iterates must agree to the bit."""
    import re
    common = {"HYP_SYRK_TAIL": "0", "HYP_SYRK_S": "3", "HYP_CHOL_SPLIT_S": "3", "HYP_CHOL_SPLIT_MIN_N": "1024", "HYP_CHOL_SPLIT_STATS": "1"}
    one = _run("psd_split", dict(common, HYP_CHOL_SPLIT="0"), True)
    two = _run("psd_split", dict(common, HYP_CHOL_SPLIT="60"), True)
    plain = _run("psd_split", dict(common, HYP_CHOL_SPLIT="45", HYP_CHOL_SPLIT_FREE="0", HYP_CHOL_SPLIT_LA="0"), True)
    count = lambda r: int(re.search(r"\[chol split\] (\d+) factorizations", r["stderr"]).group(1))
    assert count(one) == 0 and count(two) >= one["iters"] and count(plain) >= one["iters"]   # (the comparison is not vacuous)
    assert one["status"] == two["status"] == plain["status"] == "Optimal"
    assert one["trace"] == two["trace"] == plain["trace"]


@pytest.mark.parametrize("name", ["psd_single_wide", "psd_wide_plan", "polymin_large_dual"])
def test_point_upload_behind_the_queued_assembly_changes_no_bit(name):
    """round 6, HYP_UPLOAD_AFTER (default on; step_directions, combined.jl:64-121): the point and the residuals -- read by the right-hand
    sides, not by the Schur assembly or the factorization -- are staged and uploaded behind those in the stream's queue instead of in
    front of them.  The same data in the same buffers before their first reader: the same iterates to the last bit"""
    after = _run(name, {"HYP_UPLOAD_AFTER": "1"})
    front = _run(name, {"HYP_UPLOAD_AFTER": "0"})
    assert after["status"] == front["status"] == "Optimal"
    assert after["iters"] == front["iters"]
    assert after["trace"] == front["trace"], name
    assert after["trials"] == front["trials"]


@pytest.mark.parametrize("name,expect_used", [("psd_single_wide", True), ("psd_wide_plan", True), ("psd_run", True), ("polymin_large_dual", False),
                                               ("matrixcompletion", False)])
def test_cone_products_queued_at_accept_time_change_no_bit(name, expect_used):
    """round 6, HYP_SHP_PRELAUNCH=1 (measured: no gain outside the profiler, default off): the square-root Hessian products of the NEXT update_lhs (qrchol.jl:219-233) are queued the
    moment the line search accepts a candidate -- the cones then hold the state that assembly starts from -- and assemble_lhs continues
    with the Schur product if no cone has taken a point or been reset since.  The same kernels on the same data, earlier in the queue:
    the same iterates to the last bit; models with a cone that does not go through its square root never prelaunch."""
    import re
    on = _run(name, {"HYP_SHP_PRELAUNCH": "1", "HYP_SHP_PRELAUNCH_STATS": "1"}, True)
    off = _run(name, {"HYP_SHP_PRELAUNCH": "0", "HYP_SHP_PRELAUNCH_STATS": "1"}, True)
    m = re.search(r"\[sqrt_hess_prod prelaunch\] used (\d+), not used (\d+)", on["stderr"])
    used, unused = int(m.group(1)), int(m.group(2))
    if expect_used:
        assert used >= on["iters"] - 2 and unused <= 1, (used, unused, on["iters"])   # (every iteration but the first; the last one's is never asked for)
    else:
        assert used == 0
    assert re.search(r"used 0, not used 0", off["stderr"])
    assert on["status"] == off["status"] == "Optimal"
    assert on["iters"] == off["iters"]
    assert on["trace"] == off["trace"], name
    assert on["trials"] == off["trials"]
