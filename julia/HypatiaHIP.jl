# HypatiaHIP.jl -- the reference-side binding of libhypatia_hip.so (include/hypatia_hip.h).
#
# New subtypes of Hypatia's two extension points and nothing else:
#   * Cones.Cone{Float64}                      (src/Cones/Cones.jl:27)          -> fourteen HIP cone types below
#   * Solvers.QRCholSystemSolver{Float64}      (systemsolvers/qrchol.jl:14)     -> HIPQRCholDenseSystemSolver
#   * Solvers.SymIndefSystemSolver{Float64}    (systemsolvers/symindef.jl:29)   -> HIPSymIndefDenseSystemSolver
# Use:  solver = Solvers.Solver{Float64}(syssolver = HypatiaHIP.HIPQRCholDenseSystemSolver())
#       with a Models.Model whose cones are HypatiaHIP.PosSemidefTri(dim), HypatiaHIP.EpiNormSpectral(d1, d2), ...
#
# Julia is not installed in the image this repository is built in: the file has been written against Hypatia v0.5.1's
# sources (citations are file:line of that tree) and reviewed by hand, not executed.  The Python mirror hypatia.jl_amd/
# binds the same symbols with the same argument order (hypatia.jl_amd/_lib.py SIGNATURES, checked against the header by
# tests/test_capi_symbols.py) and is what the test-suite drives; tests/c/abi_smoke.c drives them from plain C.
#
# One line of Hypatia itself has to learn about the new Nonnegative type: rescale_data (src/Solvers/process.jl:37) tests
# `cone isa Cones.Nonnegative` to give every ROW of a nonnegative cone its own h-scale.  `is_nonnegative` below is the
# trait to test instead (`cone isa Cones.Nonnegative || HypatiaHIP.is_nonnegative(cone)`); without that change a model
# with HIP Nonnegative cones is rescaled with one h-scale per cone (valid, but a different scaling from the CPU path).
module HypatiaHIP

import Hypatia
import Hypatia.Cones
import Hypatia.Models
import Hypatia.Solvers
using LinearAlgebra

const lib = get(ENV, "HYPATIA_HIP_LIB", joinpath(@__DIR__, "..", "hypatia.jl_amd", "libhypatia_hip.so"))

const CTX = Ref{Ptr{Cvoid}}(C_NULL)

last_error() = unsafe_string(ccall((:hyp_last_error, lib), Cstring, (Ptr{Cvoid},), CTX[]))

function check(rc::Integer, what::AbstractString)
    rc == 0 || error(what, " failed (", rc, "): ", last_error())
    return
end

function __init__()
    ndev = Ref{Cint}(0)
    check(ccall((:hyp_device_count, lib), Cint, (Ptr{Cint},), ndev), "hyp_device_count")
    ndev[] > 0 || error("no HIP device visible: HypatiaHIP needs an MI355X (there is no CPU fallback)")
    dev = parse(Int, get(ENV, "LOCAL_RANK", "0")) % Int(ndev[])
    check(ccall((:hyp_ctx_create, lib), Cint, (Cint, Ptr{Ptr{Cvoid}}), dev, CTX), "hyp_ctx_create")
    atexit(() -> ccall((:hyp_ctx_destroy, lib), Cint, (Ptr{Cvoid},), CTX[]))
    return
end

# =============================================================================================
# cones
# =============================================================================================
# Host-side fields = the ones Hypatia's driver reads directly: cone.point, cone.dual_point, cone.vec1, cone.vec2
# (steppers/common.jl:37-46, 96-105), cone.grad / cone.dder3 as return buffers, cone.dim / cone.nu
# (Cones.jl:34, 41), cone.use_dual_barrier (Cones.jl:138).  Everything else lives on the device behind `handle`.
abstract type HIPCone <: Cones.Cone{Float64} end

is_nonnegative(::Cones.Cone) = false

macro hipcone(name)
    return esc(quote
        mutable struct $name <: HIPCone
            handle::Ptr{Cvoid}
            dim::Int
            nu::Float64
            use_dual_barrier::Bool
            point::Vector{Float64}
            dual_point::Vector{Float64}
            grad::Vector{Float64}
            dder3::Vector{Float64}
            vec1::Vector{Float64}
            vec2::Vector{Float64}
            grad_host_valid::Bool
            use_hess_prod_slow::Bool
            use_hess_prod_slow_updated::Bool
            keepalive::Any     # host arrays the constructor passed by pointer (the library copies them; kept for clarity)
            $name(handle::Ptr{Cvoid}, keepalive = nothing) = finish_cone!(new(handle), keepalive)
        end
    end)
end

# dimension, nu and use_dual_barrier are read back from the device object, so that the inverted flag of
# WSOSInterpNonnegative / WSOSInterpPosSemidefTri (use_dual_barrier = !use_dual: wsosinterpnonnegative.jl:58,
# wsosinterppossemideftri.jl:62) and every cone's nu have ONE definition
function finish_cone!(cone::HIPCone, keepalive)
    d = Ref{Cint}(0)
    check(ccall((:hyp_cone_dimension, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), cone.handle, d), "hyp_cone_dimension")
    nu = Ref{Cdouble}(0.0)
    check(ccall((:hyp_cone_get_nu, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), cone.handle, nu), "hyp_cone_get_nu")
    udb = Ref{Cint}(0)
    check(ccall((:hyp_cone_use_dual_barrier, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), cone.handle, udb), "hyp_cone_use_dual_barrier")
    cone.dim = Int(d[])
    cone.nu = nu[]
    cone.use_dual_barrier = (udb[] != 0)
    cone.grad_host_valid = false
    cone.use_hess_prod_slow = false
    cone.use_hess_prod_slow_updated = false
    cone.keepalive = keepalive
    finalizer(c -> (c.handle == C_NULL || ccall((:hyp_cone_destroy, lib), Cint, (Ptr{Cvoid},), c.handle); c.handle = C_NULL), cone)
    return cone
end

@hipcone Nonnegative                # Cones.Nonnegative{Float64}                       nonnegative.jl:8-33
@hipcone PosSemidefTri              # Cones.PosSemidefTri{Float64, Float64}            possemideftri.jl:9-46
@hipcone PosSemidefTriComplex       # Cones.PosSemidefTri{Float64, ComplexF64}         possemideftri.jl:9-46 (dim = side^2)
@hipcone EpiNormSpectral            # Cones.EpiNormSpectral{Float64, Float64}          epinormspectral.jl:13-66
@hipcone EpiNormSpectralComplex     # Cones.EpiNormSpectral{Float64, ComplexF64}       epinormspectral.jl:13-66 (dim = 1 + 2 d1 d2)
@hipcone WSOSInterpNonnegative      # Cones.WSOSInterpNonnegative{Float64, Float64}    wsosinterpnonnegative.jl:16-63
@hipcone WSOSInterpNonnegativeComplex   # Cones.WSOSInterpNonnegative{Float64, ComplexF64} wsosinterpnonnegative.jl:15-63 (complex Ps, real cone vector)
@hipcone LinMatrixIneq              # Cones.LinMatrixIneq{Float64} (dense real symmetric or complex Hermitian members) linmatrixineq.jl:9-65
@hipcone DoublyNonnegativeTri       # Cones.DoublyNonnegativeTri{Float64}              doublynonnegativetri.jl:9-52
@hipcone HypoRootdetTri             # Cones.HypoRootdetTri{Float64, Float64}           hyporootdettri.jl:9-59
@hipcone HypoPerLogdetTri           # Cones.HypoPerLogdetTri{Float64, Float64}         hypoperlogdettri.jl:9-58
@hipcone HypoRootdetTriComplex      # Cones.HypoRootdetTri{Float64, ComplexF64}        hyporootdettri.jl:9-59 (dim = 1 + side^2)
@hipcone HypoPerLogdetTriComplex    # Cones.HypoPerLogdetTri{Float64, ComplexF64}      hypoperlogdettri.jl:9-58 (dim = 2 + side^2)
@hipcone WSOSInterpPosSemidefTri    # Cones.WSOSInterpPosSemidefTri{Float64}           wsosinterppossemideftri.jl:9-69

is_nonnegative(::Nonnegative) = true

new_handle() = Ref{Ptr{Cvoid}}(C_NULL)

function Nonnegative(dim::Int)
    h = new_handle()
    check(ccall((:hyp_cone_create_nonnegative, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), CTX[], dim, h), "hyp_cone_create_nonnegative")
    return Nonnegative(h[])
end

function PosSemidefTri(dim::Int)
    h = new_handle()
    check(ccall((:hyp_cone_create_possemideftri, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), CTX[], dim, h), "hyp_cone_create_possemideftri")
    return PosSemidefTri(h[])
end

function PosSemidefTriComplex(dim::Int)
    h = new_handle()
    check(ccall((:hyp_cone_create_possemideftri_complex, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), CTX[], dim, h),
        "hyp_cone_create_possemideftri_complex")
    return PosSemidefTriComplex(h[])
end

function EpiNormSpectral(d1::Int, d2::Int; use_dual::Bool = false)
    @assert 1 <= d1 <= d2                                                         # epinormspectral.jl:58
    h = new_handle()
    check(ccall((:hyp_cone_create_epinormspectral, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        CTX[], d1, d2, use_dual, h), "hyp_cone_create_epinormspectral")
    return EpiNormSpectral(h[])
end

function EpiNormSpectralComplex(d1::Int, d2::Int; use_dual::Bool = false)
    @assert 1 <= d1 <= d2                                                         # epinormspectral.jl:58
    h = new_handle()
    check(ccall((:hyp_cone_create_epinormspectral_complex, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        CTX[], d1, d2, use_dual, h), "hyp_cone_create_epinormspectral_complex")
    return EpiNormSpectralComplex(h[])
end

# Ps[k] is U x L_k, column-major: passed as an array of K pointers
function pointer_array(Ps::Vector{Matrix{Float64}})
    return Ptr{Float64}[pointer(Pk) for Pk in Ps]
end

function WSOSInterpNonnegative(U::Int, Ps::Vector{Matrix{Float64}}; use_dual::Bool = false)
    for Pk in Ps
        @assert size(Pk, 1) == U                                                  # wsosinterpnonnegative.jl:54-56
    end
    Ls = Cint[size(Pk, 2) for Pk in Ps]
    ptrs = pointer_array(Ps)
    h = new_handle()
    GC.@preserve Ps ptrs begin
        check(ccall((:hyp_cone_create_wsosinterpnonnegative, lib), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}, Ptr{Ptr{Float64}}, Cint, Ptr{Ptr{Cvoid}}),
            CTX[], U, length(Ps), Ls, ptrs, use_dual, h), "hyp_cone_create_wsosinterpnonnegative")
    end
    return WSOSInterpNonnegative(h[], Ps)
end

# complex bases (src/PolyUtils/complex.jl:13-72): a Matrix{ComplexF64} is (re, im) interleaved, column-major, which is what the
# C-ABI takes
function WSOSInterpNonnegativeComplex(U::Int, Ps::Vector{Matrix{ComplexF64}}; use_dual::Bool = false)
    for Pk in Ps
        @assert size(Pk, 1) == U                                                  # wsosinterpnonnegative.jl:54-56
    end
    Ls = Cint[size(Pk, 2) for Pk in Ps]
    ptrs = Ptr{Float64}[Ptr{Float64}(pointer(Pk)) for Pk in Ps]
    h = new_handle()
    GC.@preserve Ps ptrs begin
        check(ccall((:hyp_cone_create_wsosinterpnonnegative_complex, lib), Cint,
            (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}, Ptr{Ptr{Float64}}, Cint, Ptr{Ptr{Cvoid}}),
            CTX[], U, length(Ps), Ls, ptrs, use_dual, h), "hyp_cone_create_wsosinterpnonnegative_complex")
    end
    return WSOSInterpNonnegativeComplex(h[], Ps)
end

function WSOSInterpPosSemidefTri(R::Int, U::Int, Ps::Vector{Matrix{Float64}}; use_dual::Bool = false)
    for Pk in Ps
        @assert size(Pk, 1) == U                                                  # wsosinterppossemideftri.jl:53-55
    end
    Ls = Cint[size(Pk, 2) for Pk in Ps]
    ptrs = pointer_array(Ps)
    h = new_handle()
    GC.@preserve Ps ptrs begin
        check(ccall((:hyp_cone_create_wsosinterppossemideftri, lib), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cint}, Ptr{Ptr{Float64}}, Cint, Ptr{Ptr{Cvoid}}),
            CTX[], R, U, length(Ps), Ls, ptrs, use_dual, h), "hyp_cone_create_wsosinterppossemideftri")
    end
    return WSOSInterpPosSemidefTri(h[], Ps)
end

function LinMatrixIneq(As::Vector; use_dual::Bool = false)
    dim = length(As)
    @assert dim > 1                                                               # linmatrixineq.jl:42
    side = 0
    for A_i in As
        if A_i isa AbstractMatrix
            side = iszero(side) ? size(A_i, 1) : side
            @assert size(A_i, 1) == side
        end
        @assert ishermitian(A_i)
    end
    @assert side > 0
    @assert Cones.svec_length(side) >= dim
    @assert isposdef(first(As))
    h = new_handle()
    if any(A_i -> A_i isa AbstractMatrix && eltype(A_i) <: Complex, As)             # complex Hermitian members: the cone vector stays real
        cstacked = zeros(ComplexF64, side, side, dim)                             # (re, im) interleaved, column-major
        for (i, A_i) in enumerate(As)
            cstacked[:, :, i] .= (A_i isa UniformScaling ? Matrix{ComplexF64}(A_i, side, side) : Matrix{ComplexF64}(A_i))
        end
        check(ccall((:hyp_cone_create_linmatrixineq_complex, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{ComplexF64}, Cint, Ptr{Ptr{Cvoid}}),
            CTX[], dim, side, cstacked, use_dual, h), "hyp_cone_create_linmatrixineq_complex")
        return LinMatrixIneq(h[])
    end
    stacked = zeros(side, side, dim)                                              # member i at stacked[:, :, i], column-major
    for (i, A_i) in enumerate(As)
        stacked[:, :, i] .= (A_i isa UniformScaling ? Matrix{Float64}(A_i, side, side) : Matrix{Float64}(A_i))
    end
    check(ccall((:hyp_cone_create_linmatrixineq, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Ptr{Cvoid}}),
        CTX[], dim, side, stacked, use_dual, h), "hyp_cone_create_linmatrixineq")
    return LinMatrixIneq(h[])
end

for (T, sym) in ((:DoublyNonnegativeTri, :hyp_cone_create_doublynonnegativetri),
                 (:HypoRootdetTri, :hyp_cone_create_hyporootdettri),
                 (:HypoPerLogdetTri, :hyp_cone_create_hypoperlogdettri),
                 (:HypoRootdetTriComplex, :hyp_cone_create_hyporootdettri_complex),
                 (:HypoPerLogdetTriComplex, :hyp_cone_create_hypoperlogdettri_complex))
    @eval function $T(dim::Int; use_dual::Bool = false)
        h = new_handle()
        check(ccall(($(QuoteNode(sym)), lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}), CTX[], dim, use_dual, h), $(string(sym)))
        return $T(h[])
    end
end

# ---- the protocol of Cones.jl:27-310, every oracle forwarded -----------------------------------
Cones.dimension(cone::HIPCone) = cone.dim                                         # Cones.jl:34
Cones.get_nu(cone::HIPCone) = cone.nu                                             # Cones.jl:41
Cones.use_dual_barrier(cone::HIPCone) = cone.use_dual_barrier                     # Cones.jl:138 (value read from the device object)
Cones.use_dder3(::HIPCone) = true                                                 # Cones.jl:126

function Cones.setup_data!(cone::HIPCone)                                         # Cones.jl:140-153
    d = cone.dim
    cone.point = zeros(d)
    cone.dual_point = zeros(d)
    cone.grad = zeros(d)
    cone.dder3 = zeros(d)
    cone.vec1 = zeros(d)
    cone.vec2 = zeros(d)
    Cones.reset_data(cone)
    return cone
end

function Cones.reset_data(cone::HIPCone)                                          # Cones.jl:185-186
    cone.grad_host_valid = false
    cone.use_hess_prod_slow = false
    cone.use_hess_prod_slow_updated = false
    check(ccall((:hyp_cone_reset_data, lib), Cint, (Ptr{Cvoid},), cone.handle), "hyp_cone_reset_data")
    return
end

function Cones.set_initial_point!(arr::AbstractVector, cone::HIPCone)
    tmp = zeros(cone.dim)
    check(ccall((:hyp_cone_set_initial_point, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), cone.handle, tmp), "hyp_cone_set_initial_point")
    copyto!(arr, tmp)
    return arr
end

function Cones.load_point(cone::HIPCone, point::AbstractVector{Float64}, scal::Float64)   # Cones.jl:157-162
    @. cone.point = scal * point
    p = Vector{Float64}(point)       # (views of the solver's Point are strided: pass a contiguous copy)
    check(ccall((:hyp_cone_load_point, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cdouble), cone.handle, p, scal), "hyp_cone_load_point")
    cone.grad_host_valid = false
    return cone.point
end

function Cones.load_point(cone::HIPCone, point::AbstractVector)                   # Cones.jl:164-166
    copyto!(cone.point, point)
    check(ccall((:hyp_cone_load_point, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cdouble), cone.handle, cone.point, 1.0), "hyp_cone_load_point")
    cone.grad_host_valid = false
    return cone.point
end

function Cones.load_dual_point(cone::HIPCone, point::AbstractVector)              # Cones.jl:168-171
    copyto!(cone.dual_point, point)
    check(ccall((:hyp_cone_load_dual_point, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), cone.handle, cone.dual_point), "hyp_cone_load_dual_point")
    return cone.dual_point
end

function device_flag(sym::Symbol, cone::HIPCone)
    out = Ref{Cint}(0)
    # (ccall needs a literal function name: one branch per boolean oracle)
    rc = if sym === :is_feas
        ccall((:hyp_cone_is_feas, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), cone.handle, out)
    elseif sym === :is_dual_feas
        ccall((:hyp_cone_is_dual_feas, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), cone.handle, out)
    elseif sym === :check_numerics
        ccall((:hyp_cone_check_numerics, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), cone.handle, out)
    else
        ccall((:hyp_cone_update_use_hess_prod_slow, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), cone.handle, out)
    end
    check(rc, string(sym))
    return (out[] != 0)
end

Cones.is_feas(cone::HIPCone) = device_flag(:is_feas, cone)                        # Cones.jl:56
Cones.is_dual_feas(cone::HIPCone) = device_flag(:is_dual_feas, cone)              # Cones.jl:63
# (called with the default tolerances only: search.jl:127; they are the library's)
Cones.check_numerics(cone::HIPCone) = device_flag(:check_numerics, cone)          # Cones.jl:273-290

function Cones.grad(cone::HIPCone)                                                # Cones.jl:71
    if !cone.grad_host_valid
        check(ccall((:hyp_cone_grad, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), cone.handle, cone.grad), "hyp_cone_grad")
        cone.grad_host_valid = true
    end
    return cone.grad
end

# products: prod / arr are dim x ncols with their own leading dimensions (SubArray views of HGQ2, qrchol.jl:162-165)
for (jl, sym) in ((:hess_prod!, :hyp_cone_hess_prod), (:inv_hess_prod!, :hyp_cone_inv_hess_prod),
                  (:sqrt_hess_prod!, :hyp_cone_sqrt_hess_prod), (:inv_sqrt_hess_prod!, :hyp_cone_inv_sqrt_hess_prod),
                  (:hess_prod_slow!, :hyp_cone_hess_prod_slow))
    @eval function Cones.$jl(prod::StridedVecOrMat{Float64}, arr::StridedVecOrMat{Float64}, cone::HIPCone)
        @assert stride(prod, 1) == 1 && stride(arr, 1) == 1
        ldp = (prod isa AbstractVector) ? length(prod) : stride(prod, 2)
        lda = (arr isa AbstractVector) ? length(arr) : stride(arr, 2)
        GC.@preserve prod arr begin
            check(ccall(($(QuoteNode(sym)), lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint),
                cone.handle, pointer(prod), ldp, pointer(arr), lda, size(arr, 2)), $(string(sym)))
        end
        return prod
    end
end

function Cones.use_sqrt_hess_oracles(arr_dim::Int, cone::HIPCone)                 # Cones.jl:189-195
    out = Ref{Cint}(0)
    check(ccall((:hyp_cone_use_sqrt_hess_oracles, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}), cone.handle, arr_dim, out),
        "hyp_cone_use_sqrt_hess_oracles")
    return (out[] != 0)
end

function Cones.update_use_hess_prod_slow(cone::HIPCone)                           # Cones.jl:222-231
    cone.use_hess_prod_slow = device_flag(:update_use_hess_prod_slow, cone)
    cone.use_hess_prod_slow_updated = true
    return
end

# test/cone.jl:89-95 sets the field by hand; setproperty! keeps the device's copy of the switch in step
function Base.setproperty!(cone::HIPCone, name::Symbol, value)
    if name === :use_hess_prod_slow && isdefined(cone, :handle) && getfield(cone, :handle) != C_NULL
        check(ccall((:hyp_cone_set_use_hess_prod_slow, lib), Cint, (Ptr{Cvoid}, Cint), getfield(cone, :handle), Bool(value)),
            "hyp_cone_set_use_hess_prod_slow")
    end
    return setfield!(cone, name, convert(fieldtype(typeof(cone), name), value))
end

function Cones.dder3(cone::HIPCone, dir::AbstractVector)                          # Cones.jl:134
    d = Vector{Float64}(dir)
    check(ccall((:hyp_cone_dder3, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), cone.handle, d, cone.dder3), "hyp_cone_dder3")
    return cone.dder3
end

function Cones.get_proxsqr(cone::HIPCone, irtmu::Float64, use_max_prox::Bool)     # Cones.jl:294-310, nonnegative.jl:137-145
    out = Ref{Cdouble}(0.0)
    check(ccall((:hyp_cone_get_proxsqr, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint, Ptr{Cdouble}), cone.handle, irtmu, use_max_prox, out),
        "hyp_cone_get_proxsqr")
    return out[]
end

# explicit Hessians (tests, SymIndef / Naive system solvers): dim x dim, upper triangle meaningful (Cones.jl:79-93)
function Cones.hess(cone::HIPCone)
    H = zeros(cone.dim, cone.dim)
    check(ccall((:hyp_cone_hess, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), cone.handle, H), "hyp_cone_hess")
    return Symmetric(H, :U)
end

function Cones.inv_hess(cone::HIPCone)
    H = zeros(cone.dim, cone.dim)
    check(ccall((:hyp_cone_inv_hess, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), cone.handle, H), "hyp_cone_inv_hess")
    return Symmetric(H, :U)
end

# dense Hessians: the generic sparsity queries of Cones.jl:262-270 apply unchanged

# =============================================================================================
# QRCholDenseSystemSolver (systemsolvers/qrchol.jl:104-257)
# =============================================================================================
# load / update_lhs / solve_subsystem3 are the three methods a QRCholSystemSolver subtype defines; the shared reductions
# solve_system / solve_subsystem4 (common.jl:129-182) and setup_rhs3 (qrchol.jl:16-37) are inherited and call the cones'
# hess_prod! / inv_hess_prod! above.  Every cone of the model must be a HIPCone: the Schur assembly reads device state.
mutable struct HIPQRCholDenseSystemSolver <: Solvers.QRCholSystemSolver{Float64}
    handle::Ptr{Cvoid}
    rhs_sub::Solvers.Point{Float64}
    sol_sub::Solvers.Point{Float64}
    sol_const::Solvers.Point{Float64}
    rhs_const::Solvers.Point{Float64}
    use_sqrt_hess_cones::Vector{Cint}
    last_info::Int
    fallback_kind::Int       # 0 Cholesky, 1 Bunch-Kaufman (rook), 2 increase_diag! + Bunch-Kaufman (dense.jl:194-215)
    HIPQRCholDenseSystemSolver() = (s = new(); s.handle = C_NULL; s)
end

function cone_handles(model::Models.Model{Float64})
    all(c -> c isa HIPCone, model.cones) || error("the HIP system solvers need every cone of the model to be a HypatiaHIP cone")
    return Ptr{Cvoid}[c.handle for c in model.cones]
end

function Solvers.load(sys::HIPQRCholDenseSystemSolver, solver::Solvers.Solver{Float64})   # qrchol.jl:138-179
    model = solver.model
    (n, p, q) = (model.n, model.p, model.q)
    handles = cone_handles(model)
    h = new_handle()
    check(ccall((:hyp_sys_create, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}),
        CTX[], n, p, q, handles, length(handles), h), "hyp_sys_create")
    sys.handle = h[]
    finalizer(Solvers.free_memory, sys)
    G = Matrix{Float64}(model.G)
    if iszero(p)
        check(ccall((:hyp_sys_load, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            sys.handle, G, C_NULL, C_NULL, C_NULL, C_NULL), "hyp_sys_load")
    else
        # G * Ap_Q (qrchol.jl:154) is formed on the device from the explicit Q factor and Ap_R
        Q = Matrix{Float64}(solver.Ap_Q * Matrix{Float64}(I, n, n))
        R = Matrix{Float64}(solver.Ap_R)
        check(ccall((:hyp_sys_load, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            sys.handle, G, C_NULL, C_NULL, Q, R), "hyp_sys_load")
    end
    # the model vectors for the device-resident direction solves (optional entry points, INTEGRATION.md)
    A = iszero(p) ? C_NULL : Matrix{Float64}(model.A)
    check(ccall((:hyp_sys_load_model, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        sys.handle, Vector{Float64}(model.c), Vector{Float64}(model.b), Vector{Float64}(model.h), A), "hyp_sys_load_model")
    sys.use_sqrt_hess_cones = zeros(Cint, length(handles))
    sys.last_info = 0
    sys.fallback_kind = 0
    Solvers.setup_point_sub(sys, model)                                           # common.jl:184-208
    return sys
end

function Solvers.update_lhs(sys::HIPQRCholDenseSystemSolver, solver::Solvers.Solver{Float64})   # qrchol.jl:181-199
    model = solver.model
    info = Ref{Cint}(0)
    fb = Ref{Cint}(0)
    if model.n > model.p                                                          # isempty(Q2div) || update_lhs_fact
        solver.time_upfact += @elapsed check(ccall((:hyp_sys_update_lhs_fact, lib), Cint,
            (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}), sys.handle, sys.use_sqrt_hess_cones, info, fb), "hyp_sys_update_lhs_fact")
        sys.last_info = Int(info[])
        sys.fallback_kind = Int(fb[])
        iszero(info[]) || println("positive definite linear system factorization failed")   # qrchol.jl:252-254
    end
    # rhs_const.z_k = H_k h_k (block_hess_prod!, qrchol.jl:191-195), all cones in one call
    check(ccall((:hyp_sys_block_hess_prod, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        sys.handle, sys.rhs_const.z, Vector{Float64}(model.h)), "hyp_sys_block_hess_prod")
    Solvers.solve_subsystem3(sys, solver, sys.sol_const, sys.rhs_const)
    return sys
end

function Solvers.solve_subsystem3(sys::HIPQRCholDenseSystemSolver, solver::Solvers.Solver{Float64},
    sol::Solvers.Point{Float64}, rhs::Solvers.Point{Float64})                     # qrchol.jl:39-85
    check(ccall((:hyp_sys_solve3, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), sys.handle, sol.vec, rhs.vec), "hyp_sys_solve3")
    return sol
end

function Solvers.free_memory(sys::HIPQRCholDenseSystemSolver)                     # Solvers.jl:407, 582
    if sys.handle != C_NULL
        ccall((:hyp_sys_destroy, lib), Cint, (Ptr{Cvoid},), sys.handle)
        sys.handle = C_NULL
    end
    return
end

# =============================================================================================
# SymIndefDenseSystemSolver (systemsolvers/symindef.jl:203-271)
# =============================================================================================
# setup_rhs3 (symindef.jl:33-56) and the 6 -> 4 -> 3 reductions are inherited; used with reduce = false
# (test/runnativetests.jl:80-86, 101-118).
mutable struct HIPSymIndefDenseSystemSolver <: Solvers.SymIndefSystemSolver{Float64}
    handle::Ptr{Cvoid}
    rhs_sub::Solvers.Point{Float64}
    sol_sub::Solvers.Point{Float64}
    sol_const::Solvers.Point{Float64}
    rhs_const::Solvers.Point{Float64}
    last_info::Int
    HIPSymIndefDenseSystemSolver() = (s = new(); s.handle = C_NULL; s)
end

function Solvers.load(sys::HIPSymIndefDenseSystemSolver, solver::Solvers.Solver{Float64})   # symindef.jl:218-237
    model = solver.model
    (n, p, q) = (model.n, model.p, model.q)
    handles = cone_handles(model)
    h = new_handle()
    check(ccall((:hyp_symindef_create, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}),
        CTX[], n, p, q, handles, length(handles), h), "hyp_symindef_create")
    sys.handle = h[]
    finalizer(Solvers.free_memory, sys)
    A = iszero(p) ? C_NULL : Matrix{Float64}(model.A)
    check(ccall((:hyp_symindef_load, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), sys.handle, A, Matrix{Float64}(model.G)),
        "hyp_symindef_load")
    sys.last_info = 0
    Solvers.setup_point_sub(sys, model)
    return sys
end

function Solvers.update_lhs(sys::HIPSymIndefDenseSystemSolver, solver::Solvers.Solver{Float64})   # symindef.jl:239-261
    info = Ref{Cint}(0)
    fb = Ref{Cint}(0)
    # z-blocks from the cones' explicit (inverse) Hessians + symm_fact_copy! (dense.jl:170-184) on the device
    solver.time_upfact += @elapsed check(ccall((:hyp_symindef_update_lhs, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}),
        sys.handle, info, fb), "hyp_symindef_update_lhs")
    sys.last_info = Int(info[])
    iszero(info[]) || println("symmetric linear system factorization failed")     # symindef.jl:254-256
    Solvers.solve_subsystem3(sys, solver, sys.sol_const, sys.rhs_const)
    return sys
end

function Solvers.solve_subsystem3(sys::HIPSymIndefDenseSystemSolver, ::Solvers.Solver{Float64},
    sol::Solvers.Point{Float64}, rhs::Solvers.Point{Float64})                     # symindef.jl:263-271
    copyto!(sol.vec, rhs.vec)
    check(ccall((:hyp_symindef_solve3, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), sys.handle, sol.vec, rhs.vec), "hyp_symindef_solve3")
    return sol
end

function Solvers.free_memory(sys::HIPSymIndefDenseSystemSolver)
    if sys.handle != C_NULL
        ccall((:hyp_symindef_destroy, lib), Cint, (Ptr{Cvoid},), sys.handle)
        sys.handle = C_NULL
    end
    return
end

# =============================================================================================
# products with the device-resident model.G (apply_lhs, calc_convergence_params: common.jl:91-94, Solvers.jl:432, 450)
# =============================================================================================
# y = alpha * op(G) * x + beta * y without moving G: optional, the CPU products with model.G stay valid
function mul_G!(y::Vector{Float64}, sys::HIPQRCholDenseSystemSolver, trans::Bool, x::Vector{Float64}, alpha::Float64 = 1.0, beta::Float64 = 0.0)
    check(ccall((:hyp_sys_mul_G, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ptr{Float64}, Cdouble, Ptr{Float64}),
        sys.handle, trans, alpha, x, beta, y), "hyp_sys_mul_G")
    return y
end

# G' z, G x + s, h' z and z' s of one point from ONE pass over the device-resident G: what calc_convergence_params
# (Solvers.jl:432, 450, 468-472) forms with two products; on a cone-sharded solver z and s are the rank's rows and the x-space
# results are summed over the ranks
function residual_products!(Gtz::Vector{Float64}, Gx_s::Vector{Float64}, dots::Vector{Float64}, sys::HIPQRCholDenseSystemSolver,
        x::Vector{Float64}, z::Vector{Float64}, s::Vector{Float64})
    check(ccall((:hyp_sys_residual_products, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        sys.handle, x, z, s, Gtz, Gx_s, dots), "hyp_sys_residual_products")
    return (Gtz, Gx_s, dots)
end

# the same with the two residual norms of Solvers.jl:447-457 over ALL ranks' rows (norms = [max |G x + s|, max |G x + s - h tau|]) riding
# in the same exchange: a cone-sharded host needs no collective of its own per iteration
function residual_products_norms!(Gtz::Vector{Float64}, Gx_s::Vector{Float64}, dots::Vector{Float64}, norms::Vector{Float64},
        sys::HIPQRCholDenseSystemSolver, x::Vector{Float64}, z::Vector{Float64}, s::Vector{Float64}, tau::Float64)
    check(ccall((:hyp_sys_residual_products2, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        sys.handle, x, z, s, tau, Gtz, Gx_s, dots, norms), "hyp_sys_residual_products2")
    return (Gtz, Gx_s, dots, norms)
end

# rank / world of the transport behind hyp_sys_set_comm's callback (an MPI communicator's, say): with it the scalars of a solve travel
# behind its n-vectors in one all-reduce; hyp_sys_set_comm_rccl takes the layout from its communicator
function set_comm_layout!(sys::HIPQRCholDenseSystemSolver, rank::Integer, world::Integer)
    check(ccall((:hyp_sys_set_comm_layout, lib), Cint, (Ptr{Cvoid}, Cint, Cint), sys.handle, rank, world), "hyp_sys_set_comm_layout")
    return sys
end

# exchanges issued since creation, by place in the iteration (see hyp_sys_comm_hist in include/hypatia_hip.h)
function comm_hist(sys::HIPQRCholDenseSystemSolver)
    out = zeros(Clonglong, 16)
    check(ccall((:hyp_sys_comm_hist, lib), Cint, (Ptr{Cvoid}, Ptr{Clonglong}), sys.handle, out), "hyp_sys_comm_hist")
    return out
end

# milliseconds spent in those exchanges, by the same places ([15] = slot 14 of the C array: the Schur exchange with its pack / unpack)
function comm_times(sys::HIPQRCholDenseSystemSolver)
    out = zeros(Cdouble, 16)
    check(ccall((:hyp_sys_comm_times, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), sys.handle, out), "hyp_sys_comm_times")
    return out
end

# solve plans built since the context was created, and how many of them run one refinement step instead of two (adaptive rule)
function plan_stats()
    out = zeros(Clonglong, 2)
    check(ccall((:hyp_ctx_plan_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Clonglong}), CTX[], out), "hyp_ctx_plan_stats")
    return out
end

# fall-backs behind a failed Cholesky since the context was created: [hybrid, calls trimmed by the growth guard, plain rook pivoting]
function bk_stats()
    out = zeros(Clonglong, 3)
    check(ccall((:hyp_ctx_bk_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Clonglong}), CTX[], out), "hyp_ctx_bk_stats")
    return out
end

# ---------------------------------------------------------------------------------------------
# The fused fast path of step(::CombinedStepper) (steppers/combined.jl:53-120), for a stepper method that wants it (p = 0):
#   step_directions!      update_lhs + update_rhs_cent / _pred / _centadj / _predadj + the two paired solves in ONE device call
#   search_alpha_resident the schedule walk of search.jl:46-69 on the point and directions that call left on the device: for models
#                         of equal PosSemidefTri cones all remaining candidates are formed there and screened side by side, only the
#                         survivor goes through check_cone_points; returns the accepted alpha (0 = none) with the accepted
#                         candidate's z / tau / s / kap rows in cand_ztsk (copy them into point.ztsk: they are what the cones hold)
# ---------------------------------------------------------------------------------------------
function step_directions!(dirs4::Matrix{Float64}, res_norms::Vector{Float64}, sys::HIPQRCholDenseSystemSolver, solver, residuals::Vector{Float64})
    ns = Ref{Cint}(0); info = Ref{Cint}(0); fb = Ref{Cint}(0)
    flags = zeros(Cint, max(length(solver.model.cones), 1))
    check(ccall((:hyp_sys_step_directions, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint, Cdouble, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Cint}, Ptr{Cint},
            Ptr{Cint}, Ptr{Cint}, Ptr{Float64}),
        sys.handle, solver.point.vec, residuals, solver.tau_residual, solver.mu, solver.max_ref_steps, solver.res_norm_cutoff, 0.5, dirs4,
        res_norms, ns, flags, info, fb, sys.sol_const.vec), "hyp_sys_step_directions")
    return (info[] == 0, Int(ns[]))
end

# Only the x rows and tau / kap of the four directions are copied into dirs4 by step_directions! (x_rows_only = true): all a stepper
# needs on the host when it walks the schedule with search_alpha_resident (the accepted candidate's z / tau / s / kap rows come back
# from there; update_stepper_points, steppers/combined.jl:124-170, then forms the x rows only)
function set_direction_rows!(sys::HIPQRCholDenseSystemSolver, x_rows_only::Bool)
    check(ccall((:hyp_sys_set_direction_rows, lib), Cint, (Ptr{Cvoid}, Cint), sys.handle, x_rows_only ? 1 : 0), "hyp_sys_set_direction_rows")
    return sys
end

function search_screen_usable(sys::HIPQRCholDenseSystemSolver)
    usable = Ref{Cint}(0); screens = Ref{Clonglong}(0); rejected = Ref{Clonglong}(0)
    check(ccall((:hyp_sys_search_screen_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Clonglong}, Ptr{Clonglong}),
        sys.handle, usable, screens, rejected), "hyp_sys_search_screen_stats")
    return usable[] != 0
end

function search_alpha_resident(cand_ztsk::Vector{Float64}, sys::HIPQRCholDenseSystemSolver, stepper, searcher, start_sched::Int)
    sched = Vector{Float64}(searcher.alpha_sched)
    idx = Ref{Cint}(-1); prox = Ref{Cdouble}(0.0); nt = Ref{Cint}(0); nl = Ref{Cint}(0); irtmu = Ref{Cdouble}(0.0)
    check(ccall((:hyp_sys_search_alpha_resident, lib), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Cint, Cint, Cdouble, Cdouble, Cint, Cdouble, Ptr{Float64}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint},
            Ptr{Cint}, Ptr{Cdouble}),
        sys.handle, stepper.unadj_only, stepper.cent_only, sched, length(sched), start_sched - 1, searcher.min_prox, searcher.prox_bound,
        searcher.use_max_prox, searcher.nup1, cand_ztsk, idx, prox, nt, nl, irtmu), "hyp_sys_search_alpha_resident")
    if idx[] >= 0
        searcher.prox = prox[]
        searcher.prev_sched = idx[] + 1
        return sched[idx[] + 1]
    end
    searcher.prev_sched = length(sched) + 1
    return 0.0
end

# =============================================================================================
# preprocessing: column-pivoted QR of [A; G] on the device (find_initial_x, src/Solvers/process.jl:64-178)
# =============================================================================================
# dgeqp3's semantics (what qr!(AG, ColumnNorm()) calls): returns (p, R, qtb) with AG[:, p] = Q R and qtb = Q' rhs, so that the
# body of find_initial_x reads  AG_rank = count(abs(R[i, i]) > tol);  init_x[p] = R \ qtb[1:n]  unchanged.
function qr_pivoted_device(AG::Matrix{Float64}, rhs::Vector{Float64})
    (m, n) = size(AG)
    h = new_handle()
    check(ccall((:hyp_qrcp_factor, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Ptr{Cvoid}}),
        CTX[], m, n, AG, m, rhs, h), "hyp_qrcp_factor")
    r = min(m, n)
    jpvt = zeros(Cint, n)
    R = zeros(r, n)
    qtb = zeros(m)
    check(ccall((:hyp_qrcp_get, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        h[], jpvt, R, C_NULL, qtb), "hyp_qrcp_get")
    check(ccall((:hyp_qrcp_destroy, lib), Cint, (Ptr{Cvoid},), h[]), "hyp_qrcp_destroy")
    return (Int.(jpvt) .+ 1, UpperTriangular(R[:, 1:r]), qtb)
end

end # module
