"""GPU: the cone-sharded path with the HIP back end, 2 ranks sharing the one GPU of the test box
(gloo transport on CUDA tensors stands in for RCCL, which needs one device per rank)."""
import multiprocessing as mp
import os
import tempfile

import numpy as np
import pytest

from test_distributed_gloo import _free_port

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_sharded_hip_solve_with_runs_of_equal_cones(monkeypatch):
    """8 equal PSD cones, 4 per rank: each rank's cones form a run (group arena, batched feasibility / inverses / products /
    proximity scalars) inside the native sharded step"""
    monkeypatch.setenv("HYP_DIST_NATIVE", "1")
    _run_sharded("1", inst_args=(90, [6] * 8, 3))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("native", ["1", "0"])
def test_sharded_hip_solve_matches_oracle(native, monkeypatch):
    """native = 1: every rank runs the fused device routines on its rows / cones and the library calls back for the
    all-reduces (hyp_sys_set_comm); native = 0: the host-composed distributed driver."""
    monkeypatch.setenv("HYP_DIST_NATIVE", native)
    _run_sharded(native)


def _run_sharded(native, inst_args=(120, [20, 12, 16, 9], 4)):
    import dist_worker
    from oracle import instances as I
    from oracle.build import make_model
    from oracle.solvers import Solver as OSolver
    port = _free_port()
    out = os.path.join(tempfile.mkdtemp(), "dist_hip.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dist_worker.run, args=(r, 2, port, inst_args, out, "hip")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = np.load(out)
    ref = OSolver(verbose=False)
    ref.load(make_model(I.psd_blocks(*inst_args)))
    ref.solve()
    assert bool(res["hooked"]) == (native == "1")
    assert str(res["status"]) == ref.status == "Optimal"
    assert abs(int(res["iters"]) - ref.num_iters) <= 1
    assert abs(float(res["p_obj"]) - ref.primal_obj) <= 1e-7 * (1 + abs(ref.primal_obj))
    assert np.allclose(res["x"], ref.get_x(), rtol=1e-5, atol=1e-7)
