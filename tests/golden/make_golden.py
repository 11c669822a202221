"""Generate the fixtures of tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

Two kinds of data, both small:
  known_answers.json   the expected outcomes the REFERENCE's own tests hold for the deterministic instances of
                       test/nativeinstances.jl (status / objective / solution entries, with the file:line they come from),
                       exported from oracle/instances.py so that they exist as data next to the tests;
  cone_vectors.npz     seeded inputs and the CPU oracle's outputs for every cone oracle of the hot path (grad,
                       hess_prod, inv_hess_prod, sqrt_hess_prod, dder3, proximity) at small sizes;
  trajectory_psd.json  the oracle's iterate trajectory (objective, mu, tau, step sizes) on a small PSD instance;
  trajectory_{ens,wsos,mixed}.json   the same for matrix completion / EpiNormSpectral models, polymin in both forms /
                       WSOSInterpNonnegative models and mixed PSD + spectral + WSOS models (full solves, and the first three
                       iterations of config 3b at 50 x 100 and of polymin at U = 680), each with the oracle's trajectory on the
                       1-ulp-perturbed model (three draws for the full solves) and, per
                       iterate, how far the acceptance tests of the third-order terms were from their threshold beside it.
The reference (pure Julia) cannot be executed here, so these vectors are produced by the oracle restatement, which is
itself pinned by the reference's identities and known answers (tests/test_oracle_*.py).  They freeze the oracle: a
later edit of oracle/ that changes any number fails tests/test_golden.py, and the GPU tests compare against the same
committed numbers."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import instances as I          # noqa: E402
from oracle.build import make_cone          # noqa: E402
from oracle import polyutils as pu          # noqa: E402

CONE_CASES = {
    "nonnegative_6": ("nonnegative", 6),
    "possemideftri_side5": ("possemideftri", 15),
    "possemideftri_side19": ("possemideftri", 190),
    "epinormspectral_3x5": ("epinormspectral", 3, 5, False),
    "epinormspectral_2x4_dual": ("epinormspectral", 2, 4, True),
}


def _lmi_members(side, count, seed):   # test/cone.jl:280-289 (rand_herms, real members)
    rng = np.random.default_rng(seed)
    Ah = rng.standard_normal((side, side))
    As = [Ah @ Ah.T + np.eye(side)]
    for _ in range(count - 1):
        M = rng.standard_normal((side, side))
        As.append(np.triu(M) + np.triu(M, 1).T)
    return [0.5 * (A + A.T) for A in As]


CONE_CASES["linmatrixineq_side4_dim5"] = ("linmatrixineq", _lmi_members(4, 5, 7), False)
CONE_CASES["doublynonnegativetri_side6"] = ("doublynonnegativetri", 21, False)
CONE_CASES["hyporootdettri_side5"] = ("hyporootdettri", 16, False)
CONE_CASES["hypoperlogdettri_side5_dual"] = ("hypoperlogdettri", 17, True)

# seeds of the oracle points: fixed per case, so that adding a case leaves the committed vectors of the others unchanged
SEEDS = {"epinormspectral_2x4_dual": 100, "epinormspectral_3x5": 101, "nonnegative_6": 102, "possemideftri_side19": 103,
         "possemideftri_side5": 104, "wsos_2var_halfdeg3": 105, "wsos_2var_halfdeg3_dual": 106, "linmatrixineq_side4_dim5": 107,
         "doublynonnegativetri_side6": 108, "hyporootdettri_side5": 109,
         "hypoperlogdettri_side5_dual": 110, "wsospsd_2var_halfdeg2_R2": 111}


def wsos_spec(use_dual):
    rng = np.random.default_rng(5)
    U, pts, Ps = pu.interpolate_box([-1.0, -1.0], [1.0, 1.0], 3, rng=rng, sample_factor=10)
    return ("wsosinterpnonnegative", U, Ps, use_dual)


def wsospsd_spec():
    U, pts, Ps = pu.interpolate_box([-1.0, -1.0], [1.0, 1.0], 2, sample=False)
    return ("wsosinterppossemideftri", 2, U, Ps, False)


def cone_vectors(spec, seed):
    """a perturbed interior primal / dual point pair and every oracle output at it"""
    rng = np.random.default_rng(seed)
    cone = make_cone(spec)
    dim = cone.dimension()
    cone.setup_data(); cone.reset_data()
    pt = np.zeros(dim)
    cone.set_initial_point(pt)
    cone.load_point(pt)
    assert cone.is_feas()
    dual = -np.array(cone.get_grad())
    pt = pt + 0.05 * (2 * rng.random(dim) - 1) * (1 + np.abs(pt))
    dual = dual + 0.05 * (2 * rng.random(dim) - 1) * (1 + np.abs(dual))
    cone.reset_data()
    cone.load_point(pt); cone.load_dual_point(dual)
    assert cone.is_feas() and cone.is_dual_feas()
    out = {"point": pt.copy(), "dual_point": dual.copy(), "grad": np.array(cone.get_grad())}
    arr = np.asfortranarray(rng.standard_normal((dim, 3)))
    out["arr"] = arr.copy()
    for name in ("hess_prod", "inv_hess_prod"):
        prod = np.zeros_like(arr)
        getattr(cone, name)(prod, arr)
        out[name] = prod
    if cone.use_sqrt_hess_oracles(3):
        for name in ("sqrt_hess_prod", "inv_sqrt_hess_prod"):
            prod = np.zeros_like(arr)
            getattr(cone, name)(prod, arr)
            out[name] = prod
    d = rng.standard_normal(dim)
    out["dder3_dir"] = d.copy()
    out["dder3"] = np.array(cone.dder3(d))
    out["proxsqr_max"] = np.array([cone.get_proxsqr(1.0, True)])
    out["proxsqr_sum"] = np.array([cone.get_proxsqr(1.0, False)])
    return out


def main():
    # ---- the reference's known answers as data
    ka = {}
    for name, fn in sorted(I.KNOWN_ANSWER.items()):
        inst = fn()
        exp = inst[6]
        rec = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in exp.items() if not callable(v)}
        rec["source"] = "test/nativeinstances.jl, instance %s (line range cited at its generator in oracle/instances.py)" % name.split("_")[0]
        rec["n"], rec["p"], rec["q"] = int(len(inst[0])), int(len(inst[2])), int(len(inst[4]))
        ka[name] = rec
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)

    # ---- cone oracle vectors
    cases = dict(CONE_CASES)
    flat = {}
    specs = {k: v for k, v in cases.items()}
    specs["wsos_2var_halfdeg3"] = wsos_spec(False)
    specs["wsos_2var_halfdeg3_dual"] = wsos_spec(True)
    specs["wsospsd_2var_halfdeg2_R2"] = wsospsd_spec()
    for name, spec in sorted(specs.items()):
        vec = cone_vectors(spec, seed=SEEDS[name])
        for k, v in vec.items():
            flat[name + "/" + k] = v
    np.savez_compressed(os.path.join(HERE, "cone_vectors.npz"), **flat)

    # ---- oracle trajectory on a small PSD instance
    from oracle.build import make_model
    from oracle.solvers import Solver
    inst = I.psd_blocks(40, [7, 5], seed=11)
    rows = []
    s = Solver()
    s.iter_callback = lambda sv: rows.append([sv.primal_obj, sv.dual_obj, sv.mu, sv.point.tau, getattr(sv.stepper, "prev_alpha", 1.0)])
    s.load(make_model(inst)); s.solve()
    with open(os.path.join(HERE, "trajectory_psd.json"), "w") as f:
        json.dump({"instance": "oracle.instances.psd_blocks(40, [7, 5], seed=11)", "status": s.status, "num_iters": s.num_iters,
                   "columns": ["primal_obj", "dual_obj", "mu", "tau", "alpha"], "rows": rows,
                   "x": s.get_x().tolist()}, f, indent=1)
    write_trajectories()
    print("wrote", sorted(os.listdir(HERE)))


# trajectories of models over the other two cones of the hot path and of a mixed model (tests/trajectory_harness.py names the
# instances).  Each record also holds the trajectory of the SAME oracle on a model whose G and h were moved by one ulp per
# entry: the prefix on which the two agree in every step size is where a comparison of step sizes means something.
TRAJECTORIES = {
    "trajectory_ens.json": [("mc_6x8_2", {}), ("mc_12x20_3", {}), ("epinormspectral2_primal", {}), ("epinormspectral3_3x4_dual", {}),
                            ("mc_50x100_1", {"iter_limit": 3})],
    "trajectory_wsos.json": [("polymin_2_3_p_2", {}), ("polymin_2_3_d_2", {}), ("polymin_3_4_p_3", {}), ("wsosinterpnonnegative2", {}),
                             ("polymin_3_7_p_3", {"iter_limit": 3}), ("polymin_3_7_d_3", {"iter_limit": 3})],
    "trajectory_mixed.json": [("mixed_psd_ens_wsos", {}), ("mixed_dual_barriers", {}), ("mixed_two_wsos_two_ens", {})],
}


def write_trajectories():
    os.environ["HYP_GOLDEN_REGEN"] = "1"    # (instances are rebuilt from scratch, not from the recorded point choices)
    import trajectory_harness as T
    from oracle import polyutils
    for fname, cases in TRAJECTORIES.items():
        out = {"columns": list(T.COLS), "cases": {}}
        for name, opts in cases:
            polyutils.LAST_KEEP = None
            inst = T.instance(name)
            keep = polyutils.LAST_KEEP
            o = T.oracle_trajectory(inst, **opts)
            ps = [T.oracle_trajectory(T.perturbed(inst, seed=99 + j), **opts) for j in range(1 if opts else 3)]
            out["cases"][name] = {"opts": opts, "status": o["status"], "num_iters": o["iters"], "primal_obj": o["p_obj"],
                                  "rows": o["rows"].tolist(), "perturbed_rows": [p["rows"].tolist() for p in ps],
                                  "gate_decades": [None if not np.isfinite(g) else round(float(g), 3) for g in o["gate"]]}
            if name.startswith("polymin_") and keep is not None:   # (the pivoted QR's choice of points: BLAS-build dependent)
                out["cases"][name]["interp_keep"] = [int(v) for v in keep]
        with open(os.path.join(HERE, fname), "w") as f:
            json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
