"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/hypatia_hip.h declares;
the product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hypatia_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hyp_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    import hypatia_jl_amd as H
    lib = H._lib.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), "missing symbol " + s
    # every declared function is bound in the ctypes table too (keeps the Python mirror in sync)
    bound = set(H._lib.SIGNATURES) | {"hyp_last_error"}
    assert set(syms) <= bound, sorted(set(syms) - bound)


def test_header_declares_every_function_once():
    """no duplicated prototypes, and every prototype directly follows its own doc block or a sibling prototype"""
    txt = open(os.path.join(ROOT, "include", "hypatia_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:int|const char\*)\s+(hyp_[A-Za-z0-9_]+)\s*\(", txt, flags=re.M)
    dups = sorted({n for n in names if names.count(n) > 1})
    assert not dups, "declared more than once: %s" % dups
    assert len(names) == len(_declared_symbols())


def test_no_cpu_fallback_without_gpu():
    import hypatia_jl_amd as H
    import ctypes
    lib = H._lib.load_library()
    n = ctypes.c_int(-1)
    lib.hyp_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(H._lib.HypatiaHipError):
        H.PosSemidefTri(6)
    with pytest.raises(H._lib.HypatiaHipError):
        H.Nonnegative(3)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "hypatia.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def _declared_prototypes():
    """{name: number of parameters} parsed from include/hypatia_hip.h"""
    txt = open(os.path.join(ROOT, "include", "hypatia_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(hyp_[A-Za-z0-9_]+)\s*\(", txt):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth:                       # matching parenthesis of the parameter list (function-pointer parameters nest)
            depth += {"(": 1, ")": -1}.get(txt[j], 0)
            j += 1
        params = txt[i:j - 1].strip()
        if params in ("", "void"):
            out[name] = 0
            continue
        n, depth = 1, 0
        for ch in params:
            depth += {"(": 1, ")": -1}.get(ch, 0)
            if ch == "," and depth == 0:
                n += 1
        out[name] = n
    return out


def test_ctypes_signatures_match_the_header():
    """the ctypes table of the Python mirror declares as many parameters as the C prototypes (ABI drift guard)"""
    import hypatia_jl_amd as H
    protos = _declared_prototypes()
    for name, argtypes in H._lib.SIGNATURES.items():
        assert name in protos, name + " is bound but not declared in include/hypatia_hip.h"
        assert len(argtypes) == protos[name], (name, len(argtypes), protos[name])


def _header_prototypes():
    txt = open(os.path.join(ROOT, "include", "hypatia_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:int|const char\*)\s+(hyp_[A-Za-z0-9_]+)\s*\((.*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        # a function-pointer parameter contains its own parentheses and commas: count top-level commas only
        depth, n = 0, (0 if args in ("", "void") else 1)
        for ch in args:
            depth += ch == "("
            depth -= ch == ")"
            n += (ch == "," and depth == 0)
        protos[m.group(1)] = n
    return protos


def test_julia_binding_matches_the_header():
    """julia/HypatiaHIP.jl (the reference-side binding; Julia is not installed here, so it cannot be run): every ccall names a
    declared symbol and passes as many arguments as the prototype has, the argument-type tuple and the argument list agree
    in length, and all cone constructors and both system solvers of the boundary are bound."""
    src = open(os.path.join(ROOT, "julia", "HypatiaHIP.jl")).read()
    src = re.sub(r"#.*", "", src)
    protos = _header_prototypes()
    seen = set()
    # ccall((:sym, lib), Ret, (T1, T2, ...), a1, a2, ...)  -- balanced-parenthesis scan from each "ccall("
    for m in re.finditer(r"ccall\(\(\s*(:hyp_[A-Za-z0-9_]+|\$\(QuoteNode\((?:sym|c)\)\))\s*,\s*lib\)\s*,", src):
        i = m.end()
        depth, parts, cur = 1, [], ""
        while depth > 0:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                parts.append(cur.strip()); cur = ""
            else:
                cur += ch
            i += 1
        parts.append(cur.strip())
        ret, types, args = parts[0], parts[1], parts[2:]
        assert ret in ("Cint", "Cstring"), ret
        inner = types.strip()[1:-1].strip()
        depth, ntypes = 0, (1 if inner.rstrip(",") else 0)
        for ch in inner.rstrip(","):
            depth += ch in "({"
            depth -= ch in ")}"
            ntypes += (ch == "," and depth == 0)
        assert ntypes == len(args), (m.group(1), ntypes, args)
        if m.group(1).startswith(":"):
            name = m.group(1)[1:]
            assert name in protos, "julia binds an undeclared symbol " + name
            assert protos[name] == ntypes, (name, protos[name], ntypes)
            seen.add(name)
    # the @eval-generated families
    for name in ("hyp_cone_hess_prod", "hyp_cone_inv_hess_prod", "hyp_cone_sqrt_hess_prod", "hyp_cone_inv_sqrt_hess_prod", "hyp_cone_hess_prod_slow",
                 "hyp_cone_create_doublynonnegativetri", "hyp_cone_create_hyporootdettri", "hyp_cone_create_hypoperlogdettri",
                 "hyp_cone_create_hyporootdettri_complex", "hyp_cone_create_hypoperlogdettri_complex"):
        assert ":" + name in src and name in protos
        seen.add(name)
    must = {"hyp_cone_create_nonnegative", "hyp_cone_create_possemideftri", "hyp_cone_create_epinormspectral", "hyp_cone_create_wsosinterpnonnegative",
            "hyp_cone_create_linmatrixineq", "hyp_cone_create_doublynonnegativetri", "hyp_cone_create_hyporootdettri", "hyp_cone_create_hypoperlogdettri",
            "hyp_cone_create_wsosinterppossemideftri", "hyp_cone_create_possemideftri_complex", "hyp_cone_create_epinormspectral_complex",
            "hyp_cone_create_linmatrixineq_complex", "hyp_cone_create_hyporootdettri_complex", "hyp_cone_create_hypoperlogdettri_complex", "hyp_cone_create_wsosinterpnonnegative_complex", "hyp_cone_use_dual_barrier", "hyp_cone_get_nu", "hyp_cone_dimension", "hyp_sys_create",
            "hyp_sys_load", "hyp_sys_update_lhs_fact", "hyp_sys_solve3", "hyp_sys_block_hess_prod", "hyp_symindef_create", "hyp_symindef_load",
            "hyp_symindef_update_lhs", "hyp_symindef_solve3", "hyp_cone_update_use_hess_prod_slow", "hyp_cone_set_use_hess_prod_slow"}
    assert must <= seen, sorted(must - seen)
    # no hard-coded use_dual_barrier: WSOSInterpNonnegative inverts use_dual (wsosinterpnonnegative.jl:58)
    assert "use_dual_barrier(::HIPCone) = false" not in src
    # begin / end balance (a cheap check of the block structure): comprehension `for`s open nothing, `abstract type` does
    flat = re.sub(r'"[^"\n]*"', '""', src)
    flat = re.sub(r"\[[^\[\]\n]*\bfor\b[^\[\]\n]*\]", "[]", flat)
    opens = len(re.findall(r"\b(?:function|if|for|while|begin|quote|let|macro|module|struct|do|abstract type)\b", flat))
    ends = len(re.findall(r"\bend\b", flat))
    assert opens == ends, (opens, ends)
