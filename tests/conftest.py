"""pytest configuration.

* ``-m "not gpu"``: oracle-vs-golden, host logic, C-ABI export checks and the kernels executed in
  the SIMT emulator (tests/simt_emu: same .hip sources compiled for the host).
* ``-m gpu``: the parity tests proper, through libet_hip.so on a real MI355X.
The fixture ``hip`` yields the torch device the kernels run on for the selected mode.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def emu_lib_path():
    from tests.simt_emu import build as emu_build
    return emu_build.build()


@pytest.fixture(scope="session")
def hip_lib_path():
    from efficientteacher_amd.csrc import build as hip_build
    return hip_build.build()


class _Mode:
    def __init__(self, device, emulated):
        self.device = torch.device(device)
        self.emulated = emulated

    def t(self, a, dtype=None):
        x = torch.as_tensor(np.ascontiguousarray(a)) if not isinstance(a, torch.Tensor) else a
        if dtype is not None:
            x = x.to(dtype)
        return x.to(self.device)


@pytest.fixture
def emu(emu_lib_path):
    """Emulator only, with its race-exposing modes (tests/simt_emu/include/hip/hip_runtime.h): ``configure(dma_late, seed)``
    -- LDS-DMA landing at the LATEST legal moment (the s_waitcnt that retires it) instead of the earliest, and waves
    scheduled one at a time in a seeded random order between workgroup barriers."""
    import ctypes
    from efficientteacher_amd import _lib
    _lib._use_library_for_tests(emu_lib_path, emulated=True)
    dll = ctypes.CDLL(emu_lib_path)
    m = _Mode("cpu", True)
    m.configure = lambda dma_late, seed: dll.emu_configure(int(dma_late), int(seed))
    yield m
    dll.emu_configure(0, -1)
    _lib._use_library_for_tests(None, False)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def hip(request):
    """Run the test body once on the emulator (CPU) and once on the GPU (marked gpu)."""
    from efficientteacher_amd import _lib
    if request.param == "emu":
        path = request.getfixturevalue("emu_lib_path")
        _lib._use_library_for_tests(path, emulated=True)
        yield _Mode("cpu", True)
        _lib._use_library_for_tests(None, False)
    else:
        if not torch.cuda.is_available():
            pytest.fail("-m gpu selected but no GPU is visible")
        _lib._use_library_for_tests(None, False)
        _lib.load()  # raises loudly if libet_hip.so is missing
        yield _Mode("cuda:0", False)
