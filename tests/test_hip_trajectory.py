"""Iterate-level parity of whole solves for the cones beyond PosSemidefTri: EpiNormSpectral (matrix completion), WSOSInterpNonnegative
(polymin, both forms) and mixed PSD + spectral + WSOS models, against the committed oracle trajectories of tests/golden/
(`make_golden.py` wrote them; `trajectory_harness.py` names the instances).

What is composed here and pinned nowhere else: the right-hand-side builders with their third-order terms and acceptance tests
(steppers/common.jl:7-118), the refinement loop (systemsolvers/common.jl:15-76) and the line search's accept / reject decisions
(search.jl:74-138) ON THESE CONES, i.e. with the generic factored inverse Hessian (Cones.jl:113-118) behind the proximity test.

Two routes through the library, each in a process of its own because the switches are read once:
  reference  HYP_ENS_CLOSED_INV=0 HYP_PROX_LB=0 HYP_ENS_PREFETCH=0 HYP_WSOS_PAR=0: the reference's order of operations link by link;
  default    the closed-form spectral inverse Hessian, candidates rejected on the proximity lower bound, side-by-side evaluation.
Bar on both: same status; the same line-search step sizes on the prefix where the oracle's own trajectory survives 1-ulp
perturbations of G and h, mu >= 1e-7, and no acceptance test of a third-order term (`dder3_viol < 1e-4`, steppers/common.jl:47, 105)
has been decided within two decades of its threshold -- that test thresholds the rounding error of the cone's third-order oracle
(trajectory_harness.gate_margins), which on spectral cones reaches 1e-4 at mu ~ 1e-5, and from then on which of two correct
implementations keeps the term is not determined by the algorithm; objective / mu / tau / residual norms to 1e-10 relative while mu >= 1e-3 and within
100x the oracle's own 1-ulp sensitivity everywhere on that prefix.  A failure names the first diverging iterate."""
import json
import os

import numpy as np
import pytest

import trajectory_harness as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = ("trajectory_ens.json", "trajectory_wsos.json", "trajectory_mixed.json")


def _gold():
    out = {}
    for f in FILES:
        d = json.load(open(os.path.join(GOLD, f)))
        for name, rec in d["cases"].items():
            out[name] = rec
    return out


CASES = _gold()
SLOW_ON_CPU = {"mc_50x100_1"}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden_trajectories(name):
    """freezes the oracle: an edit of oracle/ that moves any iterate of these solves fails here (CPU)"""
    rec = CASES[name]
    if name in SLOW_ON_CPU and not os.environ.get("HYP_SLOW_TESTS"):
        pytest.skip("35 s of explicit 5001 x 5001 Hessians on the CPU; HYP_SLOW_TESTS=1 runs it")
    o = T.oracle_trajectory(T.instance(name), **rec["opts"])
    assert o["status"] == rec["status"] and o["iters"] == rec["num_iters"]
    g = np.array(rec["rows"])
    assert o["rows"].shape == g.shape
    assert np.array_equal(o["rows"][:, 8], g[:, 8])
    # early iterates to 1e-9; late ones drift with the BLAS build's summation order (~1/mu amplification)
    well = g[:, 7] >= 1e-6
    assert np.allclose(o["rows"][well], g[well], rtol=1e-8, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["reference", "default"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_trajectory_matches_oracle(name, route):
    rec = CASES[name]
    ht = T.hip_trajectory(name, T.REFERENCE_ROUTE if route == "reference" else T.DEFAULT_ROUTE, **rec["opts"])
    gate = np.array([np.inf if g is None else g for g in rec["gate_decades"]])
    ot = dict(status=rec["status"], iters=rec["num_iters"], rows=np.array(rec["rows"]), gate=gate)
    pt = dict(rows=[np.array(p) for p in rec["perturbed_rows"]])
    rep = T.compare(ht, ot, pt, label="%s/%s" % (name, route))
    # the compared prefix must reach well into the solve: every iterate with mu >= 1e-5 at least (a test that compared two
    # iterates would say nothing), or all of a truncated solve
    need = len(ot["rows"]) if rec["opts"].get("iter_limit") else int(np.sum(ot["rows"][:, 7] >= 1e-5))
    assert rep["prefix"] >= min(need, T.stable_prefix(ot["rows"], pt["rows"], gate)), rep
    assert rep["prefix"] >= 4, rep
    if not rec["opts"].get("iter_limit"):
        assert abs(ht["iters"] - rec["num_iters"]) <= (0 if rep["prefix"] >= len(ot["rows"]) else 3), (ht["iters"], rec["num_iters"])
        assert abs(ht["p_obj"] - rec["primal_obj"]) <= 1e-7 * (1 + abs(rec["primal_obj"]))
