"""Construct HIP cones / models from the (kind, params...) descriptions used by tests and bench."""
from . import cones as hc
from .models import Model


def make_cone(spec):
    kind = spec[0]
    if kind == "nonnegative":
        return hc.Nonnegative(spec[1])
    if kind == "possemideftri":
        return hc.PosSemidefTri(spec[1])
    if kind == "possemideftri_complex":
        return hc.PosSemidefTriComplex(spec[1])
    if kind == "epinormspectral":
        return hc.EpiNormSpectral(spec[1], spec[2], use_dual=spec[3])
    if kind == "epinormspectral_complex":
        return hc.EpiNormSpectralComplex(spec[1], spec[2], use_dual=spec[3])
    if kind == "wsosinterpnonnegative":
        return hc.WSOSInterpNonnegative(spec[1], spec[2], use_dual=spec[3])
    if kind == "wsosinterpnonnegative_complex":
        return hc.WSOSInterpNonnegativeComplex(spec[1], spec[2], use_dual=spec[3])
    if kind in ("linmatrixineq", "linmatrixineq_complex"):   # (complex Hermitian members are recognised by their dtype)
        return hc.LinMatrixIneq(spec[1], use_dual=spec[2])
    if kind == "doublynonnegativetri":
        return hc.DoublyNonnegativeTri(spec[1], use_dual=spec[2])
    if kind == "hyporootdettri":
        return hc.HypoRootdetTri(spec[1], use_dual=spec[2])
    if kind == "hypoperlogdettri":
        return hc.HypoPerLogdetTri(spec[1], use_dual=spec[2])
    if kind == "hyporootdettri_complex":
        return hc.HypoRootdetTriComplex(spec[1], use_dual=spec[2])
    if kind == "hypoperlogdettri_complex":
        return hc.HypoPerLogdetTriComplex(spec[1], use_dual=spec[2])
    if kind == "wsosinterppossemideftri":
        return hc.WSOSInterpPosSemidefTri(spec[1], spec[2], spec[3], use_dual=spec[4])
    raise NotImplementedError("no HIP cone for %r yet (and there is no CPU fallback)" % (kind,))


def make_model(inst):
    c, A, b, G, h, specs = inst[:6]
    offset = inst[6].get("obj_offset", 0.0) if len(inst) > 6 else 0.0
    return Model(c, A, b, G, h, [make_cone(s) for s in specs], obj_offset=offset)
