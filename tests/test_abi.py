"""The C-ABI boundary (include/et_hip.h): the gfx950 library builds, loads without a GPU, exports every declared
entry point, the ctypes table binds every one of them, and the product path fails loudly -- no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

from tests.conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "et_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|size_t)\s+(et_[a-z0-9_]+)\s*\(", src, flags=re.M)
    assert len(names) >= 30
    return sorted(set(names))


def test_library_exports_every_declared_symbol(hip_lib_path):
    from efficientteacher_amd import _lib
    lib = ctypes.CDLL(hip_lib_path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.et_build_arch.restype = ctypes.c_char_p
    assert lib.et_build_arch() == b"gfx950"
    assert lib.et_abi_version() == _lib.ABI_VERSION        # _lib._declare refuses any other library


def test_ctypes_table_binds_every_symbol():
    from efficientteacher_amd import _lib
    decl = set(_declared())
    bound = set(_lib.SIGNATURES)
    assert decl - bound == set(), sorted(decl - bound)
    assert bound - decl == set(), sorted(bound - decl)


def test_stale_library_is_refused(hip_lib_path, monkeypatch):
    """a library whose et_abi_version() differs from include/et_hip.h's ET_ABI_VERSION does not bind (a stale .so would
    read a newly added int argument as the stream and launch unordered on the null stream)"""
    from efficientteacher_amd import _lib
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.EtHipError) as e:
        _lib._declare(ctypes.CDLL(hip_lib_path))
    assert "rebuild" in str(e.value)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from efficientteacher_amd import _lib
    _lib._use_library_for_tests(None, False)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libet_hip.so"))
    monkeypatch.setattr(_lib, "_dll", None)
    with pytest.raises(_lib.EtHipError) as e:
        _lib.load()
    assert "no CPU fallback" in str(e.value)


def test_cpu_tensor_is_rejected_not_emulated():
    """Outside the test-only emulator hook a CPU tensor never reaches a kernel (there is no CPU path)."""
    from efficientteacher_amd import _lib
    _lib._use_library_for_tests(None, False)
    with pytest.raises(_lib.EtHipError):
        _lib.ptr(torch.zeros(4))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under efficientteacher_amd/ may import it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "efficientteacher_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
