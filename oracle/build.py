"""Build oracle cone objects / models from instance descriptions (test infrastructure)."""
from . import cones as oc
from .solvers import Model


def make_cone(spec):
    kind = spec[0]
    if kind == "nonnegative":
        return oc.Nonnegative(spec[1])
    if kind == "possemideftri":
        return oc.PosSemidefTri(spec[1])
    if kind == "epinormspectral":
        return oc.EpiNormSpectral(spec[1], spec[2], use_dual=spec[3])
    if kind == "wsosinterpnonnegative":
        return oc.WSOSInterpNonnegative(spec[1], spec[2], use_dual=spec[3])
    if kind == "linmatrixineq":
        return oc.LinMatrixIneq(spec[1], use_dual=spec[2])
    if kind == "doublynonnegativetri":
        return oc.DoublyNonnegativeTri(spec[1], use_dual=spec[2])
    if kind == "hyporootdettri":
        return oc.HypoRootdetTri(spec[1], use_dual=spec[2])
    if kind == "hypoperlogdettri":
        return oc.HypoPerLogdetTri(spec[1], use_dual=spec[2])
    if kind == "wsosinterppossemideftri":
        return oc.WSOSInterpPosSemidefTri(spec[1], spec[2], spec[3], use_dual=spec[4])
    if kind == "possemideftri_complex":   # the complex Hermitian variants
        from .cones_complex import PosSemidefTriComplex
        return PosSemidefTriComplex(spec[1])
    if kind == "epinormspectral_complex":
        from .cones_complex import EpiNormSpectralComplex
        return EpiNormSpectralComplex(spec[1], spec[2], use_dual=spec[3])
    if kind == "wsosinterpnonnegative_complex":
        from .cones_complex import WSOSInterpNonnegativeComplex
        return WSOSInterpNonnegativeComplex(spec[1], spec[2], use_dual=spec[3])
    if kind == "linmatrixineq_complex":
        from .cones_complex import LinMatrixIneqComplex
        return LinMatrixIneqComplex(spec[1], use_dual=spec[2])
    if kind == "hyporootdettri_complex":
        from .cones_complex import HypoRootdetTriComplex
        c = HypoRootdetTriComplex(spec[1], use_dual=spec[2])
        return c
    if kind == "hypoperlogdettri_complex":
        from .cones_complex import HypoPerLogdetTriComplex
        return HypoPerLogdetTriComplex(spec[1], use_dual=spec[2])
    raise ValueError(kind)


def make_model(inst):
    c, A, b, G, h, specs = inst[:6]
    offset = inst[6].get("obj_offset", 0.0) if len(inst) > 6 else 0.0
    return Model(c, A, b, G, h, [make_cone(s) for s in specs], obj_offset=offset)
