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
