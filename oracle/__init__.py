"""CPU oracle for the Hypatia per-IPM-iteration hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy/scipy restatement of the reference algorithm (chriscoey/Hypatia.jl
v0.5.1, Julia) for the path named in BASELINE.json: cone barrier oracles (Nonnegative,
PosSemidefTri, EpiNormSpectral, WSOSInterpNonnegative) and the QRCholDense Newton/KKT solve, plus
the callers needed to drive it (Point, Model, preprocessing, CombinedStepper, StepSearcher,
Solver.solve).  Every function cites the reference file:line it follows.

Who may use it: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- as the checker / the timed CPU baseline ("port"), never as the product path.  The
product (`hypatia.jl_amd`) never imports this package and fails loudly when its HIP library is
missing.

Pinning status.  The reference is pure Julia and no Julia toolchain exists in the authoring
container or on the GPU box, so the reference itself cannot be executed.  The oracle is pinned
against what the reference's own tests hold for this path:
  * the oracle identities of test/cone.jl:23-114 at 1e3*eps (tests/test_oracle_cones.py),
  * finite differences of the closed-form barriers test/cone.jl:342-346, 505-513, 764-768,
  * the deterministic known-answer instances of test/nativeinstances.jl (nonnegative4,
    possemideftri1/2/8/9, epinormspectral3/4, wsosinterpnonnegative1/2/3, ...) with the
    certificate checks of nativeinstances.jl:32-86 (tests/test_oracle_instances.py).
Iterate *trajectories* are not pinned by the reference (it stores none): trajectory parity of the
HIP path is measured against this restatement only.  Heavy arithmetic here is scipy's bundled
OpenBLAS/LAPACK, standing in for Julia's stdlib OpenBLAS/LAPACK (same routines: dpotrf, dpotri,
dtrsm, dsyrk, dgeqp3, dsytrf_rook, dgesdd).
"""
