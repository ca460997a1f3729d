"""Detector assembly (host-side mirror of reference models/detector/yolo.py:45-93 and the shared
parts of models/detector/yolo_ssod.py:44-118): ``Model(cfg)`` = backbone -> neck -> head with the
reference attributes (.backbone .neck .head .stride .model_type .names .inplace), ``forward(x)`` taking
the NCHW float image batch the trainers pass, and the reference's outputs:
  train: list of 3 logits tensors (B, 3, ny, nx, 85);  eval: (z (B, 25200, 85), list).
State lives in flat HBM arenas (efficientteacher_amd/flat_state.py); activations are NHWC inside.
"""
import copy
import logging

import torch
import torch.nn as nn

from ... import _lib, ops
from ...flat_state import FlatState
from ...utils.torch_utils import initialize_weights
from ..backbone import build_backbone
from ..head import build_head
from ..head.yolov5_head import Detect
from ..neck import build_neck

LOGGER = logging.getLogger(__name__)


def check_anchor_order(m):
    # reference utils/autoanchor.py:16: Detect anchors must be ordered like the strides
    a = m.anchors.prod(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da.sign() != ds.sign():
        m.anchors[:] = m.anchors.flip(0)


class Model(nn.Module):
    def __init__(self, cfg='yolov5s.yaml'):
        super().__init__()
        self.cfg = cfg
        self.backbone = build_backbone(cfg)
        self.neck = build_neck(cfg)
        self.head = build_head(cfg)
        self.names = cfg.Dataset.names
        self.inplace = self.cfg.Model.inplace
        self.model_type = 'yolov5'
        self.export = False
        self._build_extra(cfg)
        self.check_head()
        initialize_weights(self)
        self._flat = None
        self._compute_dtype = torch.float32
        # extension key (configs/defaults.py): reproducible BatchNorm statistics in the 16-bit modes (FlatState(deterministic=))
        self._deterministic = bool(getattr(cfg.Model, "deterministic_bn", False)) if hasattr(cfg, "Model") else False

    def _build_extra(self, cfg):
        pass

    # ---- reference surface -----------------------------------------------------------------------------------
    def check_head(self):
        m = self.head
        from ..head.yolov8_head import YoloV8Detect
        if isinstance(m, YoloV8Detect):              # reference yolo.py:77-81
            m.inplace = self.inplace
            self.stride = torch.Tensor(m.stride)
            m.initialize_biases()
            self.model_type = 'yolox'
            return
        if not isinstance(m, Detect):
            raise NotImplementedError
        m.inplace = self.inplace
        # the reference probes the strides with a 256x256 forward (yolo.py:72-76); for the YoloV5 graph
        # they are the products of the stride-2 convs in front of each pyramid level: 8, 16, 32
        m.stride = torch.tensor([8., 16., 32.])
        m.anchors /= m.stride.view(-1, 1, 1)
        check_anchor_order(m)
        self.stride = m.stride
        m.initialize_biases()  # only run once

    def info(self, verbose=False, img_size=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_g = sum(x.numel() for x in self.parameters() if x.requires_grad)
        LOGGER.info(f"Model summary: {len(list(self.modules()))} layers, {n_p} parameters, {n_g} gradients")

    def fuse(self):
        # eval-mode forward already folds BatchNorm into the conv epilogue on the fly (et_bn_eval_affine)
        return self

    def half(self):
        """reference callers (val.py:212, trainer.py:475) switch to fp16 for inference / checkpoints; here the fp32 master
        weights stay and the COMPUTE dtype becomes bf16 (the default performance mode; fp16 -- same MFMA rate, the reference's own
        arithmetic -- is ``set_compute_dtype(torch.float16)``)"""
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        return self.set_compute_dtype(torch.float32)

    # ---- arenas -----------------------------------------------------------------------------------------------
    def flat_state(self):
        if self._flat is None:
            raise RuntimeError("model is not on a GPU: call model.to('cuda') first (no CPU path)")
        return self._flat

    def set_compute_dtype(self, dtype):
        """torch.float32 = parity mode (exact-f32 MFMA); torch.bfloat16 = performance mode; torch.float16 = the reference's AMP
        arithmetic (v_mfma_f32_32x32x16_f16; training then needs optim.DeviceGradScaler, as the reference needs GradScaler)."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise TypeError(dtype)
        self._compute_dtype = dtype
        self.rebuild_flat()
        return self

    def set_deterministic(self, flag=True):
        """True: BatchNorm statistics of the bf16 / fp16 modes on the bit-reproducible partial-row path (fp64 finalize) instead of
        the sharded fp32 accumulators (order of hardware atomics); the fp32 parity mode is always on that path.  Rebuilds the arenas
        like set_compute_dtype (create the optimizer afterwards)."""
        self._deterministic = bool(flag)
        self.rebuild_flat()
        return self

    def rebuild_flat(self):
        for m in self.modules():
            m.__dict__.pop("_et_slot", None)
            m.__dict__.pop("_et_flat_ref", None)
        self._flat = None
        ps = list(self.parameters())
        on_dev = all(p.is_cuda for p in ps) or (_lib.is_emulated() and all(p.device.type == 'cpu' for p in ps))
        if ps and on_dev and all(p.dtype == torch.float32 for p in ps):
            req = [p.requires_grad for p in ps]
            self._flat = FlatState(self, self._compute_dtype, deterministic=getattr(self, "_deterministic", False))
            for p, r in zip(ps, req):
                p.requires_grad_(r)
                if not r:
                    p.grad = None
        return self

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.rebuild_flat()
        return self

    def __deepcopy__(self, memo):
        flat, slots = self._flat, []
        for m in self.modules():
            slots.append((m, m.__dict__.pop("_et_slot", None), m.__dict__.pop("_et_flat_ref", None)))
        self._flat = None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            self._flat = flat
            for m, s, r in slots:
                if s is not None:
                    m._et_slot = s
                if r is not None:
                    m._et_flat_ref = r
        new.rebuild_flat()
        return new

    def __reduce_ex__(self, protocol):
        from ...utils.checkpoint import reduce_model
        return reduce_model(self, protocol)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        if self._flat is not None:
            self._flat.mark_weights_changed()
        return r

    def zero_grad(self, set_to_none=False):
        if self._flat is not None:
            self._flat.zero_grad()
        else:
            super().zero_grad(set_to_none)

    # ---- forward -------------------------------------------------------------------------------------------
    def forward(self, x, augment=False, profile=False, visualize=False):
        return self._forward_once(x, profile, visualize)

    def _features(self, x):
        flat = self.flat_state()
        flat.prepare_forward(self.training)
        if not all(t.dim() == 4 and t.shape[1] <= 8 for t in (x if isinstance(x, (list, tuple)) else (x,))):
            raise ValueError("expected an NCHW image batch (B, 3, H, W), or a list of such batches of one shape")
        # uint8 batches (what the loaders deliver) are normalised inside the pack kernel: x / 255 (ssod_trainer.py:694-696)
        x8 = ops.pack_input(x, self._compute_dtype, norm_scale=getattr(self, "input_norm_scale", 255.0))
        slots = None
        # in-place P3 / P4: only a backbone that can write its C3 / C4 outputs into a destination slice (YoloV5BackBone) together
        # with a neck that lays out the concat buffers (YoloV5Neck); any other registered pairing takes the plain path
        if (hasattr(self.neck, "concat_slots") and getattr(self.backbone, "supports_concat_dst", False)
                and x8.shape[1] % 32 == 0 and x8.shape[2] % 32 == 0):
            slots = self.neck.concat_slots(x8.shape[0], x8.shape[1] // 8, x8.shape[2] // 8, x8.dtype, x8.device)
        if slots is None:
            return self.neck(self.backbone(x8))
        (b3, o3), (b4, o4) = slots
        c3, c4, c5 = self.backbone(x8, dst_c3=(b3, o3), dst_c4=(b4, o4))
        return self.neck((c3, c4, c5), cat_bufs=(b3, b4))     # UpsampleCatFn finds P3 / P4 already in place

    def _forward_once(self, x, profile=False, visualize=False):
        return self.head(self._features(x))
