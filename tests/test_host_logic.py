"""CPU-only checks of the host-side driver logic that need no device: what step() does when every link of the
factorization chain has failed (reference: qrchol.jl:253-255 warns, combined.jl:97-117 then ends in NumericalFailure)."""
import numpy as np


class _FakeSys:
    native_directions = False

    def __init__(self, info):
        self.calls = 0
        self.last_info = info

    def update_lhs(self, solver):
        self.calls += 1
        return self


class _FakeSolver:
    def __init__(self, sysv):
        self.syssolver = sysv
        self.status = "SolveCalled"
        self.time_upsys = 0.0
        self.point = None

        class M:
            p = 0
            cones = []
        self.model = M()


def test_step_reports_numerical_failure_when_the_factorization_chain_fails():
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as HS
    st = HS.CombinedStepper()
    st.rhs = st.dir = None            # (buffers of load(); not reached on this path)
    sysv = _FakeSys(info=7)
    solver = _FakeSolver(sysv)
    assert st.step(solver) is False
    assert solver.status == "NumericalFailure"
    assert sysv.calls == 1            # the failed assembly + factorizations are not run a second time
    assert st.prev_alpha == 0.0


def test_bench_blocks_do_not_depend_on_the_number_of_ranks():
    """bench.py --config 4: every cone's rows of G come from their own stream (seed, k), so that the N-rank run and the
    single-GPU run of the strong-scaling workload are the SAME instance"""
    import bench
    a = bench.gen_block(40, 6, 3, 1)
    b = bench.gen_block(40, 6, 3, 1)
    c = bench.gen_block(40, 6, 4, 1)
    assert a.shape == (21, 40) and a.flags.f_contiguous and np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(np.std(a) * np.sqrt(40) - 1.0) < 0.15
    e = bench.svec_identity(3)
    assert np.array_equal(e, np.array([1.0, 0, 1, 0, 0, 1]))
