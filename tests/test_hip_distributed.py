"""GPU: the cone-sharded path with the HIP back end, 2 ranks sharing the one GPU of the test box
(gloo transport on CUDA tensors stands in for RCCL, which needs one device per rank)."""
import multiprocessing as mp
import os
import tempfile

import numpy as np
import pytest

from test_distributed_gloo import _free_port, run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_sharded_hip_solve_with_runs_of_equal_cones(monkeypatch):
    """8 equal PSD cones, 4 per rank: each rank's cones form a run (group arena, batched feasibility / inverses / products /
    proximity scalars) inside the native sharded step"""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    res = _run_sharded("1", inst_args=(90, [6] * 8, 3))
    # the line search's candidates are screened side by side on every rank (candidates formed from the device-resident rows, two
    # small all-reduces per search make the verdicts the same on both ranks): it ran in every iteration and rejected candidates
    usable, screens, rejected = [int(v) for v in res["screen_stats"]]
    assert usable == 1 and screens >= int(res["iters"]) and rejected > 0


@pytest.mark.timeout(600)
def test_sharded_hip_solve_one_equal_cone_per_rank(monkeypatch):
    """the weak-scaling layout (bench.py --config 2w): two equal PSD cones, ONE per rank.  The candidate screen runs per rank with one
    matrix per candidate, its two all-reduces make the verdicts the same on both ranks; same solve as the oracle's"""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    res = _run_sharded("1", inst_args=(70, [14, 14], 5))
    usable, screens, rejected = [int(v) for v in res["screen_stats"]]
    assert usable == 1 and screens >= int(res["iters"]) and rejected > 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("native", ["1", "0"])
def test_sharded_hip_solve_matches_oracle(native, monkeypatch):
    """native = 1: every rank runs the fused device routines on its rows / cones and the library calls back for the
    all-reduces (hyp_sys_set_comm); native = 0: the host-composed distributed driver."""
    monkeypatch.setenv("HYP_DIST_NATIVE", native)
    _run_sharded(native)


@pytest.mark.timeout(600)
def test_library_rccl_exchanges_one_rank(monkeypatch):
    """RCCL inside the library (hyp_comm_init_rank + hyp_sys_set_comm_rccl): the test box has ONE GPU and RCCL wants one device
    per rank, so the communicator has a single rank -- every exchange point of the fused sharded step still goes through
    ncclAllReduce on the library stream (the multi-rank data flow is covered by the gloo tests with the callback)."""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    res = _run_sharded("1", world=1, transport="nccl")
    assert bool(res["rccl_in_library"])
    assert res["lib_exchanges"][0] > 50 and res["lib_exchanges"][1] >= 120 * 121 // 2


@pytest.mark.timeout(900)
def test_schur_exchange_overlapped_with_its_production_one_rank(monkeypatch):
    """round 5, HYP_DIST_OVERLAP=G (off by default: no N > 1 hardware to decide it on): the Schur product is emitted in G row groups of
    equal area and each group's part of the upper triangle is all-reduced on the helper stream while the next group is multiplied
    (SysSolver::assemble_lhs_overlapped; qrchol.jl:219-246 is the sum being exchanged).  One rank through the in-library RCCL (the
    test box has one GPU): n = 1300 = 11 tile rows in 3 groups; the solve must be the oracle's, and the exchange count says the
    grouped path ran (3 Schur exchanges per iteration instead of 1)."""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    monkeypatch.setenv("HYP_DIST_OVERLAP", "3")
    res = _run_sharded("1", inst_args=(1300, [24] * 8, 7), world=1, transport="nccl")
    assert bool(res["rccl_in_library"])
    hist = [int(v) for v in res["comm_hist"]]
    assert hist[0] >= 3 * int(res["iters"]), (hist[0], int(res["iters"]))


@pytest.mark.timeout(900)
def test_schur_exchange_overlapped_row_groups_as_schur_products_one_rank(monkeypatch):
    """round 6: at sizes where a row group has 64 tiles or more it is launched as an instance of the Schur product -- a launch position
    only for the tiles of the trapezoid that do work, in the XCD-aware order (gemm_f64_kernel.hpp: trap_tile_map), split-K with the
    cut last round, the thin last columns by the skinny kernel, consecutive groups on two lanes.  n = 3000 + 8 (24 tile rows and an
    8-column edge) in 3 groups of ~100 tiles, K = 9840: the same solve as with the single exchange (whose path the full-size fixtures
    pin to the oracle; the oracle's own solve of this size would take minutes of host time here)."""
    import dist_worker
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    inst_args = (3008, [40] * 12, 9)
    res = {}
    for groups in ("3", "0"):
        monkeypatch.setenv("HYP_DIST_OVERLAP", groups)
        out = os.path.join(tempfile.mkdtemp(), "dist_overlap_%s.npz" % groups)
        codes = run_ranks(dist_worker.run, lambda r, port: (r, 1, port, inst_args, out, "hip", "nccl"), 1, 500)
        assert all(c == 0 for c in codes), codes
        res[groups] = dict(np.load(out))
    a, b = res["3"], res["0"]
    assert bool(a["rccl_in_library"]) and bool(b["rccl_in_library"])
    assert [int(v) for v in a["comm_hist"]][0] >= 3 * int(a["iters"])          # (three Schur exchanges per iteration: the grouped path ran)
    assert [int(v) for v in b["comm_hist"]][0] < 2 * int(b["iters"]) + 2
    assert str(a["status"]) == str(b["status"]) == "Optimal"
    assert int(a["iters"]) == int(b["iters"])
    # (two roundings of the same Schur sums: the solves end within the solver's tolerance of each other, as against the oracle's)
    assert abs(float(a["p_obj"]) - float(b["p_obj"])) <= 1e-7 * (1 + abs(float(b["p_obj"])))
    assert np.allclose(a["x"], b["x"], rtol=1e-5, atol=1e-7)


def test_library_rccl_allreduce_on_a_device_buffer():
    """hyp_comm_unique_id / _init_rank / _allreduce / _destroy on a device buffer (one rank; in a fresh process, torch first:
    one HIP runtime per process)"""
    import dist_worker
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=dist_worker.run_comm_selftest)
    p.start()
    p.join(300)
    assert p.exitcode == 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("native", ["1", "0"])
def test_sharded_total_factorization_failure(native, monkeypatch):
    """the same on two ranks: both see the failed factorization of the (identical, all-reduced) Schur matrix and both end in
    NumericalFailure -- no rank raises from a solve with a factorization that does not exist, none is left waiting in a collective"""
    monkeypatch.setenv("HYP_DIST_NATIVE", native)
    monkeypatch.setenv("HYP_FORCE_FACT_FAIL", "1")
    import dist_worker
    out = os.path.join(tempfile.mkdtemp(), "dist_fail.npz")
    codes = run_ranks(dist_worker.run, lambda r, port: (r, 2, port, (60, [8, 6, 7], 4), out, "hip", "gloo"), 2, 300)
    assert all(c == 0 for c in codes), codes
    res = np.load(out)
    assert str(res["status"]) == "NumericalFailure" and int(res["iters"]) == 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("switch", ["HYP_NO_FUSED_STEP", "HYP_NO_PAIR"])
def test_sharded_hip_solve_with_the_fused_step_switched_off(switch, monkeypatch):
    """the documented switches that take the stepper off the fused device step must stay safe on the sharded
    solver: the unfused branches read full-length point.z / point.s / z_residual, so the driver may not be row-local then
    (round-3 advisor finding: directions came out silently wrong and different per rank)"""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    monkeypatch.setenv(switch, "1")
    res = _run_sharded("1", expect_row_local=False)
    assert not bool(res["row_local"])


@pytest.mark.timeout(600)
def test_sharded_step_exchanges_per_iteration(monkeypatch):
    """round 4: what a cone-sharded iteration exchanges, by place in the iteration (hyp_sys_comm_hist).  The scalars that accompany an
    n-vector travel behind it in the SAME all-reduce (sums in shared slots, maxima in a slot per rank): the residual's h'z and norm
    exchanges are gone (sites 4, 5), a screen survivor's <z, s> comes from the screen's exchange (site 9), the residual norms of
    calc_convergence_params ride with G'z / h'z / z's (site 12 = host-requested reductions: none).  Per iteration: one Schur
    triangle, three exchanges per paired solve, one for the residual products, two for the screen, one per survivor -- at most
    12 plus three per refinement solve (with HYP_DIST_FUSED=0, the form of round 3, the same solve needs more than 18)."""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    res = _run_sharded("1", inst_args=(90, [6] * 8, 3))
    hist = [int(v) for v in res["comm_hist"]]
    iters, n_solves = int(res["iters"]), int(res["n_solves"])
    assert hist[0] == iters                                   # the Schur sum
    assert hist[4] == 0 and hist[5] == 0 and hist[12] <= 1, hist     # (one host-requested reduction outside the loop)
    # the screen ran in every iteration and its survivors did not pay for their sums again (site 9 is left to the searches that
    # are not screened: the schedule's last single candidate, the unadjusted fall-back searches of steppers/combined.jl:97-118)
    assert hist[7] >= iters and hist[8] >= iters and hist[9] < hist[10], hist
    refinements = n_solves - 4 * iters
    assert refinements >= 0
    per_iter = sum(hist) / iters
    assert per_iter <= 12.0 + 3.0 * refinements / iters + 0.5, (per_iter, hist, refinements)
    monkeypatch.setenv("HYP_DIST_FUSED", "0")
    old = _run_sharded("1", inst_args=(90, [6] * 8, 3))
    hist_old = [int(v) for v in old["comm_hist"]]
    assert sum(hist_old) / int(old["iters"]) > per_iter + 5.0, (hist_old, hist)
    assert int(old["iters"]) == iters and abs(float(old["p_obj"]) - float(res["p_obj"])) <= 1e-9 * (1 + abs(float(res["p_obj"])))


def _run_sharded(native, inst_args=(120, [20, 12, 16, 9], 4), world=2, transport="gloo", expect_row_local=True):
    import dist_worker
    from oracle import instances as I
    from oracle.build import make_model
    from oracle.solvers import Solver as OSolver
    out = os.path.join(tempfile.mkdtemp(), "dist_hip.npz")
    codes = run_ranks(dist_worker.run, lambda r, port: (r, world, port, inst_args, out, "hip", transport), world, 500)
    assert all(c == 0 for c in codes), codes
    res = np.load(out)
    ref = OSolver(verbose=False)
    ref.load(make_model(I.psd_blocks(*inst_args)))
    ref.solve()
    assert bool(res["hooked"]) == (native == "1")
    if native == "1" and expect_row_local:
        # the device-resident sharded step: every rank keeps only ITS rows of z / s and of the four directions; inside the
        # iteration loop nothing of length q crosses the host-level transport (the library's own exchanges are the n x n Schur
        # sum, n-vectors and scalars: hyp_sys_comm_stats), and the solution still matches the oracle's (below)
        assert bool(res["row_local"])
        assert int(res["max_host_payload_in_loop"]) <= max(int(res["n"]), 32), (int(res["max_host_payload_in_loop"]), int(res["q"]))
    assert str(res["status"]) == ref.status == "Optimal"
    assert abs(int(res["iters"]) - ref.num_iters) <= 1
    assert abs(float(res["p_obj"]) - ref.primal_obj) <= 1e-7 * (1 + abs(ref.primal_obj))
    assert np.allclose(res["x"], ref.get_x(), rtol=1e-5, atol=1e-7)
    return res


@pytest.mark.timeout(600)
def test_kshard_single_cone_solve_matches_oracle():
    """ONE PosSemidefTri cone, model replicated on 2 ranks (sharing the GPU, gloo callback): the Schur product is split along K,
    the partial matrices are all-reduced, everything else runs replicated -- same solve as the oracle, one exchange per
    update_lhs and no other"""
    import dist_worker
    from oracle import instances as I
    from oracle.build import make_model
    from oracle.solvers import Solver as OSolver
    inst_args = (150, [30], 2)
    out = os.path.join(tempfile.mkdtemp(), "kshard_hip.npz")
    codes = run_ranks(dist_worker.run_kshard, lambda r, port: (r, 2, port, inst_args, out, "hip"), 2, 500)
    assert all(c == 0 for c in codes), codes
    res = np.load(out)
    ref = OSolver(verbose=False)
    ref.load(make_model(I.psd_blocks(*inst_args)))
    ref.solve()
    assert str(res["status"]) == ref.status == "Optimal"
    assert abs(int(res["iters"]) - ref.num_iters) <= 1
    assert abs(float(res["p_obj"]) - ref.primal_obj) <= 1e-7 * (1 + abs(ref.primal_obj))
    assert np.allclose(res["x"], ref.get_x(), rtol=1e-5, atol=1e-7)
    # one exchange of the Schur matrix's packed upper triangle per assembly (one per iteration), nothing else
    assert int(res["iters"]) <= res["exchanges"][0] <= int(res["iters"]) + 3
    assert res["exchanges"][1] == res["exchanges"][0] * (150 * 151 // 2)


@pytest.mark.timeout(600)
def test_kshard_cone_without_square_root(monkeypatch):
    """K-panel sharding when the cone has no square-root oracle: with HYP_FORCE_BK=1 the WSOS cone's Hessian factorization is the
    Bunch-Kaufman fallback, use_sqrt_hess_oracles answers false (Cones.jl:189-195) and the Schur assembly takes the hess_prod
    branch (qrchol.jl:240-246) -- every rank contracts its rows of the cone and the all-reduce sums to the whole term.  Same
    optimum as the oracle on the same model."""
    import dist_worker
    from oracle import instances as I
    from oracle.build import make_model
    from oracle.solvers import Solver as OSolver
    monkeypatch.setenv("HYP_FORCE_BK", "1")
    inst_args = ("polymin", 2, 3, False, 2)   # dual form: n = U - 1 = 27 after the reduction, one WSOS cone of dimension 28
    out = os.path.join(tempfile.mkdtemp(), "kshard_nosqrt.npz")
    codes = run_ranks(dist_worker.run_kshard, lambda r, port: (r, 2, port, inst_args, out, "hip"), 2, 500)
    assert all(c == 0 for c in codes), codes
    res = np.load(out)
    ref = OSolver(verbose=False)
    ref.load(make_model(I.polymin(2, 3, False, seed=2)))
    ref.solve()
    assert str(res["status"]) == ref.status == "Optimal"
    assert abs(float(res["p_obj"]) - ref.primal_obj) <= 1e-6 * (1 + abs(ref.primal_obj))
    assert res["exchanges"][0] >= int(res["iters"])


def _torchrun_bench(extra, nproc=2, timeout=1500):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` exactly as the driver launches the scaling runs,
    with the gloo transport (both ranks share this box's one GPU; RCCL wants one device per rank): returns the parsed JSON line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HYP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[-1])


def _check_schema(d, n_gpus, steps=None):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "parallelism" in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    if steps is not None:
        assert d["steps"] == steps
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6      # iterations/s x s/iteration


@pytest.mark.timeout(1800)
def test_bench_scaling_line_under_torchrun_two_ranks():
    """schema rehearsal for the driver's SCALE run (8-GPU nodes are not ours to launch): config 4's generator at n = 300, the 64
    cones dealt out to two ranks, three timed iterations; the line must carry the contract's fields with N = 2"""
    d = _torchrun_bench(["--nvars", "300", "--steps", "3", "--warmup", "1", "--no-secondary"])
    _check_schema(d, 2, steps=3)
    assert "64 x PosSemidefTri" in d["metric"] and d["config"]["n"] == 300
    assert d["comm"]["collectives_per_step"] > 0 if "comm" in d else True
    # round 5: the line times its exchanges (one Schur sum per iteration and the small ones, hyp_sys_comm_times) and carries every
    # rank's phase table, so that the first real N > 1 run says where a shortfall comes from
    assert d["schur_allreduce_ms"] > 0 and d["schur_exchange_ms_incl_pack"] >= d["schur_allreduce_ms"]
    assert d["small_collectives_ms_per_step"] > 0 and d["small_collectives_ms_by_site"]
    pr = d["per_rank_ms_per_step"]
    assert len(pr["rows"]) == 2 and all(len(r) == len(pr["columns"]) for r in pr["rows"])
    col = {c: i for i, c in enumerate(pr["columns"])}
    for r in pr["rows"]:
        assert r[col["cones"]] == 32 and r[col["syrk"]] > 0 and r[col["schur_exchanges"]] >= 1.0 and r[col["schur_allreduce"]] > 0


@pytest.mark.timeout(1800)
def test_bench_config5_kshard_line_under_torchrun_two_ranks():
    """BASELINE configs[4] "1 -> 8 GPU": ONE WSOS cone (U = 4845, dual form), model replicated, Schur product K-sharded over two ranks;
    the line says what scales and what is replicated, and the only exchange is one Schur triangle per iteration"""
    d = _torchrun_bench(["--config", "5d", "--steps", "3"])
    _check_schema(d, 2)
    assert d["config"]["final_status"] == "Optimal"
    ks = d["k_shard"]
    assert 1.0 <= ks["estimated_speedup_vs_1gpu_from_these_phases"] <= ks["amdahl_bound_any_n"] < 3.0
    assert "replicated" in ks and "scales_with_n_gpus" in ks
