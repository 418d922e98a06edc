"""CPU: the C-ABI shared library builds for gfx950, loads without a GPU, and exports exactly the symbols that
include/f2n_abi.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import build
    path = build.build_hip()
    return ctypes.CDLL(path)


def declared():
    txt = open(os.path.join(ROOT, "include", "f2n_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(f2n_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = declared()
    assert len(names) >= 36
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_only_queries(lib):
    assert lib.f2n_abi_version() == 7
    lib.f2n_build_info.restype = ctypes.c_char_p
    assert b"gfx950" in lib.f2n_build_info()
    assert lib.f2n_mlp_n_params(32, 64, 1) == 3072      # field MLP, SURVEY 8(a) a12
    assert lib.f2n_mlp_n_params(32, 64, 2) == 7168      # colour MLP, a13
    from oracle import capi as oc
    assert oc.mlp_n_params(32, 64, 1) == 3072 and oc.mlp_n_params(32, 64, 2) == 7168


def test_no_cpu_fallback_in_python_binding():
    """The ctypes binding refuses CPU tensors instead of silently computing elsewhere."""
    import torch
    from f2_nerf_amd import capi
    with pytest.raises(capi.F2nError):
        capi._p(torch.zeros(4), "f32")
    src = open(os.path.join(ROOT, "f2-nerf_amd", "capi.py")).read() + open(os.path.join(ROOT, "f2-nerf_amd", "__init__.py")).read()
    assert "import oracle" not in src and "from oracle" not in src
