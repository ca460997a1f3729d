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
