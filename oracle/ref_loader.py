"""Import the read-only reference tree (/root/reference) on this container's stack.

TEST INFRASTRUCTURE.  Only usable where /root/reference exists (the build
container); never imported by ``-m gpu`` tests, ``smoke()`` or ``bench.py``.
It is what ``oracle/make_golden.py`` uses to generate ``tests/golden/*.npz``.

The reference does not import unmodified on torch 2.10 / numpy 2.2 without
cv2 / torchvision / seaborn / tensorboard.  Nothing under
/root/reference is modified; the shims below live in ``sys.modules`` only:

* stub ``cv2`` (utils/general.py:23,41 only calls ``cv2.setNumThreads``),
* stub ``torchvision`` whose ``ops.nms`` is ``oracle.nms.nms_torch`` (the
  restated kernel; utils/general.py:976),
* stub ``seaborn`` / ``thop`` / ``tensorboard``,
* ``YOLOV5_CONFIG_DIR`` pointing at a scratch dir holding a copy of the font
  that utils/plots.py:64 would otherwise download,
* ``np.int`` (utils/general.py:516) and ``Tensor.clamp_`` with float-tensor
  bounds on int64 (models/assigner/yolo_anchor_assigner.py:367),
* ``Tensor.cuda`` -> identity (models/loss/loss.py:392,418 hard-code .cuda()).
"""
import os
import shutil
import sys
import tempfile
import types

REF = os.environ.get("ET_REFERENCE", "/root/reference")
_loaded = False


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Make ``import models, utils, configs, trainer`` resolve to the reference."""
    global _loaded
    if _loaded:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    import numpy as np
    import torch

    from . import nms as _nms

    # --- stubs -----------------------------------------------------------
    cv2 = _stub("cv2", setNumThreads=lambda n: None, INTER_LINEAR=1, INTER_AREA=3,
                IMREAD_COLOR=1, COLOR_BGR2RGB=4, COLOR_BGR2HSV=40, COLOR_HSV2BGR=54,
                BORDER_CONSTANT=0, FONT_HERSHEY_SIMPLEX=0, LINE_AA=16)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    tv = _stub("torchvision", __version__="0.0-stub")
    tv.ops = _stub("torchvision.ops", nms=_nms.nms_torch)
    tv.transforms = _stub("torchvision.transforms")
    tv.models = _stub("torchvision.models")
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    _stub("seaborn")
    tb = _stub("torch.utils.tensorboard")

    class SummaryWriter:  # utils/loggers/__init__.py:12
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass

    tb.SummaryWriter = SummaryWriter

    # --- font dir (utils/plots.py:64 -> check_font would hit the network) --
    cfgdir = tempfile.mkdtemp(prefix="et_ref_cfg_")
    font = os.path.join(REF, "utils", "Arial.ttf")
    if os.path.exists(font):
        shutil.copy(font, os.path.join(cfgdir, "Arial.ttf"))
    os.environ["YOLOV5_CONFIG_DIR"] = cfgdir

    # --- numpy / torch compatibility ---------------------------------------
    if not hasattr(np, "int"):
        np.int = int
    _orig_clamp_ = torch.Tensor.clamp_

    def clamp_(self, min=None, max=None):
        if not self.dtype.is_floating_point:
            if isinstance(min, torch.Tensor):
                min = int(min.item())
            if isinstance(max, torch.Tensor):
                max = int(max.item())
            if isinstance(min, float):
                min = int(min)
            if isinstance(max, float):
                max = int(max)
        return _orig_clamp_(self, min, max)

    torch.Tensor.clamp_ = clamp_
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self

    if REF not in sys.path:
        sys.path.insert(0, REF)
    _loaded = True


def get_cfg(yaml_rel=None, opts=()):
    """configs/defaults.py:325 get_cfg + merge (train.py:65-67)."""
    load()
    from configs.defaults import get_cfg as _get

    cfg = _get()
    if yaml_rel:
        cfg.merge_from_file(os.path.join(REF, yaml_rel))
    if opts:
        cfg.merge_from_list(list(opts))
    return cfg
