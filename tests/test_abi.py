"""The C-ABI library loads on a CPU-only box and exports every symbol include/pgbart.h declares
(no compute calls here: the HIP backend has no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from pymc_bart_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "pgbart.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgb_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_abi.SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    path = _abi.hip_library_path()
    assert os.path.exists(path), "run __graft_entry__.build() first"
    lib = _abi.load_hip_library()
    for name in _declared_symbols():
        assert hasattr(lib.lib, name), name
    assert lib.backend_name == "hip-gfx950"


def test_oracle_exports_the_same_abi(oracle):
    for name in _declared_symbols():
        assert hasattr(oracle.lib.lib, name), name
    assert oracle.lib.backend_name == "oracle-cpu"


def test_struct_layouts_match_the_c_side(oracle):
    lib = oracle.lib.lib
    for fn, struct in (("pgbo_sizeof_settings", _abi.Settings), ("pgbo_sizeof_counters", _abi.Counters),
                       ("pgbo_sizeof_tree_arrays", _abi.TreeArraysC)):
        f = getattr(lib, fn)
        f.restype = C.c_int64
        assert f() == C.sizeof(struct), fn


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pymc_bart_amd.sampler import default_backend

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        default_backend()


def test_product_package_never_references_the_oracle():
    pkg = os.path.join(ROOT, "pymc_bart_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libpgbart_oracle" not in text and "from _oracle" not in text, f
                assert "import oracle" not in text, f


def test_library_override_must_still_be_the_hip_backend(oracle, monkeypatch):
    """PGBART_HIP_LIB names another BUILD of the HIP library (profiling / experiment builds); pointing it
    at anything that is not the gfx950 backend -- the CPU checker, say -- is refused: the product has
    no CPU path, by override or otherwise."""
    monkeypatch.setenv("PGBART_HIP_LIB", oracle.lib.path)
    monkeypatch.setattr(_abi, "_HIP_LIB", {})
    with pytest.raises(_abi.PGBError, match="not the HIP backend"):
        _abi.load_hip_library()
    assert not _abi._HIP_LIB  # a refused library is not kept
    monkeypatch.delenv("PGBART_HIP_LIB")
    assert _abi.load_hip_library().backend_name == "hip-gfx950"


def test_both_builds_of_the_hip_library_export_the_abi_and_state_their_particle_limit():
    """libpgbart_hip.so: one particle per lane (64); libpgbart_hip_p128.so: the same source with two per lane."""
    for want in (64, 128):
        lib = _abi.load_hip_library(want)
        assert lib.backend_name == "hip-gfx950" and lib.max_particles == want
        for sym in _abi.SYMBOLS:
            assert hasattr(lib.lib, sym), (want, sym)
