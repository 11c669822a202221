"""hypatia.jl_amd -- MI355X-native hot path for Hypatia.jl's interior-point iterations.

The product is `libhypatia_hip.so` (hand-written HIP for gfx950 behind the C-ABI of
include/hypatia_hip.h).  This Python package is the host-side mirror of the reference's plugin
interface for the path -- `Cones.Cone` subtypes and `Solvers.QRCholDenseSystemSolver`
(/root/reference/src/Cones/Cones.jl:27, src/Solvers/systemsolvers/qrchol.jl:104) -- plus a mirror of
the callers (`Solvers.Solver`, `CombinedStepper`, `StepSearcher`) so the path can be driven without
Julia.  It never imports `oracle/` and has no CPU fallback: constructing a cone or a system solver
without the HIP library and a GPU raises.
"""
from . import _lib            # noqa: F401
from .cones import Nonnegative, PosSemidefTri, PosSemidefTriComplex, EpiNormSpectral, EpiNormSpectralComplex, WSOSInterpNonnegative, WSOSInterpNonnegativeComplex, LinMatrixIneq, DoublyNonnegativeTri, HypoRootdetTri, HypoPerLogdetTri, HypoRootdetTriComplex, HypoPerLogdetTriComplex, WSOSInterpPosSemidefTri, Cone   # noqa: F401
from .models import Model                                    # noqa: F401
from .systemsolvers import QRCholDenseSystemSolver, SymIndefDenseSystemSolver   # noqa: F401
from .solvers import Solver, CombinedStepper, StepSearcher, Point   # noqa: F401
from .build import make_cone, make_model                     # noqa: F401
