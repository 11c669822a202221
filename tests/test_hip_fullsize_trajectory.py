"""Whole-iteration parity AT THE BENCHMARKED SIZES: the HIP path against the committed full-size oracle trajectories
(tests/golden/trajectory_fullsize.json + schur_probe_fullsize.npz, written by tests/golden/make_golden_fullsize.py from the
instances bench.py times -- trajectory_harness.fullsize_instance).

What this puts inside an oracle-compared iteration that the small trajectory tests never select: psd_ts3_kernel on one and on two
column chunks, the 39 x 39 x 5 split-K syrk + syrk_edge, the look-ahead MFMA Cholesky, the super-block solve plan with its
refinement steps, the side-200 candidate screen (config 2 and the mid sizes); the group arena of 64 equal cones (config 4); the
U = 4845 WSOS cone with its blocked Hessian factorization, Bunch-Kaufman fallback and the dual form's trmm + syrk (config 5).

Bar = trajectory_harness.compare (north_star: "identical iterate residual norms to ~1e-10 rel"): same status, the same line-search
step sizes on the prefix where the oracle's own trajectory survives 1-ulp perturbations of G and h (mu >= 1e-7), p_obj / d_obj /
mu / tau / residual norms to 1e-10 relative while mu >= 1e-3 (3x the oracle's own 1-ulp sensitivity where that is larger: the dual form
of config 5 moves by 4e-10 under 1-ulp perturbations at mu = 1.4e-3) and within 100x the oracle's own 1-ulp sensitivity on that prefix;
the Schur matrix assembled at the initial iterate to 1e-12 of the oracle's (entries of a seeded 64 x 64 sub-block, the diagonal,
S V on probe vectors, Frobenius norm), later probes to 1e-9.  Both routes.  Reference: Solvers.jl:340-398,
steppers/combined.jl:53-120, search.jl:46-138, qrchol.jl:201-257."""
import json
import os

import numpy as np
import pytest

import trajectory_harness as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_JF = os.path.join(GOLD, "trajectory_fullsize.json")
CASES = json.load(open(_JF))["cases"] if os.path.exists(_JF) else {}


def _probes():
    z = np.load(os.path.join(GOLD, "schur_probe_fullsize.npz"))
    out = {}
    for key in z.files:
        name, it, f = key.split("/")
        out.setdefault(name, {}).setdefault(int(it), {})[f] = z[key]
    return out


def test_fullsize_fixtures_cover_the_benchmarked_configurations():
    """(CPU) the committed rows exist for config 2 as benchmarked (whole solve), two mid sizes, config 4 and config 5 in both forms"""
    for name in ("psdfull_5000_200x1_1", "psdfull_1300_113x1_1", "psdfull_2500_160x1_1", "cfg4_5000_80x64_1", "cfg5p_1", "cfg5d_1", "cfg5pw_1",
                 "cfg5dw_1"):
        assert name in CASES, name
        rec = CASES[name]
        rows = np.array(rec["rows"])
        assert rows.shape == (rec["num_iters"] + 1, len(T.COLS)) and np.all(np.isfinite(rows))
        assert all(len(p) >= 2 for p in rec["perturbed_rows"])
    assert CASES["psdfull_5000_200x1_1"]["status"] == "Optimal" and CASES["psdfull_5000_200x1_1"]["q"] == 20100
    assert CASES["cfg4_5000_80x64_1"]["q"] == 64 * 3240 and CASES["cfg5p_1"]["q"] == 4845
    # round 5: config 5 to the end in both forms (three perturbed companions for the first ten iterates), config 4 five iterations deep
    for name in ("cfg5pw_1", "cfg5dw_1"):
        assert CASES[name]["status"] == "Optimal" and not CASES[name]["opts"] and len(CASES[name]["perturbed_rows"]) == 3
        assert all(len(p) == 11 for p in CASES[name]["perturbed_rows"])
    assert CASES["cfg4_5000_80x64_1"]["num_iters"] == 5 and len(CASES["cfg4_5000_80x64_1"]["perturbed_rows"]) == 3
    P = _probes()
    assert set(P["psdfull_5000_200x1_1"]) == {1, 3} and P["psdfull_5000_200x1_1"][1]["diag"].shape == (5000,)


@pytest.mark.skipif(not os.environ.get("HYP_SLOW_TESTS"), reason="minutes of host BLAS; HYP_SLOW_TESTS=1 runs it")
@pytest.mark.parametrize("name", ["psdfull_1300_113x1_1"])
def test_oracle_reproduces_fullsize_golden(name):
    rec = CASES[name]
    o = T.oracle_trajectory(T.instance(name), **rec["opts"])
    g = np.array(rec["rows"])
    assert o["status"] == rec["status"] and o["rows"].shape == g.shape
    well = g[:, 7] >= 1e-6
    assert np.allclose(o["rows"][well], g[well], rtol=1e-8, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("route", ["default", "reference"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_fullsize_trajectory_matches_oracle(name, route):
    rec = CASES[name]
    ht = T.hip_trajectory(name, T.REFERENCE_ROUTE if route == "reference" else T.DEFAULT_ROUTE, timeout=1700,
                          probe_iters=rec["probe_iters"] if route == "default" else [], **rec["opts"])
    gate = np.array([np.inf if g is None else g for g in rec["gate_decades"]])
    ot = dict(status=rec["status"], iters=rec["num_iters"], rows=np.array(rec["rows"]), gate=gate)
    pt = dict(rows=[np.array(p) for p in rec["perturbed_rows"]])
    rep = T.compare(ht, ot, pt, label="%s/%s" % (name, route))
    truncated = bool(rec["opts"].get("iter_limit"))
    # the compared prefix: all of a truncated solve; otherwise every iterate the perturbed companions cover with mu >= 1e-5
    covered = min(len(p) for p in pt["rows"])
    need = len(ot["rows"]) if truncated else min(covered, int(np.sum(ot["rows"][:, 7] >= 1e-5)))
    assert rep["prefix"] >= min(need, T.stable_prefix(ot["rows"], pt["rows"], gate)), rep
    assert rep["prefix"] >= (3 if truncated else 5), rep
    if not truncated:
        assert abs(ht["iters"] - rec["num_iters"]) <= 3, (ht["iters"], rec["num_iters"])
        assert abs(ht["p_obj"] - rec["primal_obj"]) <= 1e-7 * (1 + abs(rec["primal_obj"]))
    if name.startswith("cfg5pw") and route == "default":   # (the dual form's Hessians never fail their Cholesky on this instance: bk_stats 0 / 0 / 0)
        # the whole solves of config 5 exist to put the fall-back behind a failed Cholesky of the cone Hessian (dense.jl:194-215) --
        # on the default route the hybrid form that keeps the Cholesky's finished block steps, csrc/bunchkaufman.hip -- inside an
        # oracle-compared solve at U = 4845: the comparison must not pass without it having run
        hybrid, trimmed, plain = ht["bk_stats"]
        assert hybrid >= 1, ht["bk_stats"]
    if route != "default":
        return
    gp = _probes().get(name, {})
    assert sorted(ht["probes"]) == sorted(gp), (sorted(ht["probes"]), sorted(gp))
    for it, g in gp.items():
        h = ht["probes"][it]
        tol = 1e-12 if it == 1 else 1e-9
        scale = g["fro"][0] / np.sqrt(g["diag"].shape[0])       # rms row norm: the entries' natural scale
        assert np.abs(h["sub"] - g["sub"]).max() <= tol * max(scale, np.abs(g["sub"]).max()), (name, it, "sub-block")
        assert np.abs(h["diag"] - g["diag"]).max() <= tol * np.abs(g["diag"]).max(), (name, it, "diagonal")
        assert np.linalg.norm(h["SV"] - g["SV"]) <= tol * np.linalg.norm(g["SV"]), (name, it, "S V")
        assert abs(h["fro"][0] - g["fro"][0]) <= tol * g["fro"][0], (name, it, "norm")


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n.startswith("cfg5")])
def test_fullsize_wsos_candidate_screen_changes_no_bit(name):
    """config 5 at U = 4845, both forms: the line search's candidate screen (csrc/wsos_screen.hip, on by default) against the
    sequential walk (HYP_WSOS_SCREEN=0) and against itself in check mode (HYP_WSOS_SCREEN_CHECK=1: every screened-out candidate
    also goes through the sequential test, a disagreement raises inside the library) -- the same iterates to the last bit and the
    same number of candidates visited"""
    rec = CASES[name]
    runs = [T.hip_trajectory(name, route, timeout=1700, **rec["opts"])
            for route in ({}, {"HYP_WSOS_SCREEN": "0"}, {"HYP_WSOS_SCREEN_CHECK": "1"})]
    on, off, chk = runs
    assert on["status"] == off["status"] == chk["status"]
    assert on["rows"].shape == off["rows"].shape == chk["rows"].shape
    assert (on["rows"] == off["rows"]).all() and (chk["rows"] == off["rows"]).all()
    assert on["trials"] == off["trials"] == chk["trials"]
    assert off["screens"] == [0, 0]
    assert on["screens"][0] > 0 and on["screens"][1] > 0      # (it ran, and it rejected something)
