"""The C-ABI from a plain C host (tests/c/abi_smoke.c): compiles and links against include/hypatia_hip.h and the in-tree
library on the CPU box (no GPU needed for that), runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
PKG = os.path.join(ROOT, "hypatia.jl_amd")


def _build(tmp_path):
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", PKG, "-l:libhypatia_hip.so", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_host_compiles_and_links_against_the_header(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_host_runs_the_path(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c abi smoke ok" in r.stdout
