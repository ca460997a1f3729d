"""Anchor-based Detect head (host-side mirror of reference models/head/yolov5_head.py:7-87,127-136)."""
import math

import torch
import torch.nn as nn

from ... import ops
from ...autograd import ConvBiasFn, head_view


class Detect(nn.Module):
    """Attribute names (`nc`, `no`, `nl`, `na`, `m`, `anchors`, `stride`, `grid`, `anchor_grid`, `export`, `cur_imgsize`,
    `num_keypoints`) are the reference's: checkpoints pickle them and `state_dict` keys (`head.m.<i>.weight`, `head.anchors`)
    depend on them (tests/test_model.py pins the key list)."""
    stride = None

    def __init__(self, cfg):
        super().__init__()
        ds, mdl = cfg.Dataset, cfg.Model
        if ds.np:
            raise NotImplementedError("keypoint heads are outside the hot path")
        self.nc, self.num_keypoints = ds.nc, ds.np
        self.cur_imgsize = [ds.img_size] * 2
        anchor_table = torch.tensor(mdl.anchors, dtype=torch.float32)              # one row of (w, h) pairs per level
        self.nl, self.na = anchor_table.shape[0], anchor_table.shape[1] // 2
        self.no = 5 + self.nc + self.num_keypoints                                   # box 4 + objectness + classes
        self.register_buffer('anchors', anchor_table.view(self.nl, self.na, 2))
        self.grid = [torch.zeros(1) for _ in range(self.nl)]
        self.anchor_grid = [torch.zeros(1) for _ in range(self.nl)]
        widths = [int(c * mdl.width_multiple) for c in mdl.Neck.out_channels]
        self.m = nn.ModuleList([nn.Conv2d(c, self.na * self.no, kernel_size=1) for c in widths])
        self.stride = mdl.Head.strides
        self.export = False

    def initialize_biases(self, cf=None):
        """Prior-probability bias init of the output convs (RetinaNet's focal-loss paper, sec. 3.3; reference
        models/head/yolov5_head.py:37-46): objectness starts at "8 objects per 640-pixel image" for the level's cell count,
        class logits at 0.6 / (nc - 0.99), or at the class frequencies `cf` when given."""
        cls_prior = math.log(0.6 / (self.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
        for conv, s in zip(self.m, self.stride):
            cells = (640 / s) ** 2
            bias = conv.bias.detach().view(self.na, self.no).clone()
            bias[:, 4] += math.log(8 / cells)
            bias[:, 5:] += cls_prior
            conv.bias = torch.nn.Parameter(bias.reshape(-1), requires_grad=True)

    def _raw(self, xi, i):
        """head conv -> logits viewed as (B, na, ny, nx, no) over the NHWC GEMM output (no copy)."""
        cs = getattr(self.m[i], "_et_slot", None)
        if cs is None:
            raise RuntimeError("model state is not on the device arenas yet (model.to('cuda')); no CPU path")
        if self.training and torch.is_grad_enabled():
            return ConvBiasFn.apply(xi, self.m[i].weight, cs, ops.ACT_NONE, (self.na, self.no))
        return head_view(ops.conv2d_fwd(xi, cs.w_lp, 1, 0, bias=cs.bias), self.na, self.no)

    def forward(self, x):
        x = list(x)
        if self.export:
            raise NotImplementedError("export path is out of scope")
        for i in range(self.nl):
            x[i] = self._raw(x[i], i)
        if self.training:
            return x
        # inference: decode every level into z (B, sum na*ny*nx, no) fp32
        B = x[0].shape[0]
        sizes = [xi.shape[1] * xi.shape[2] * xi.shape[3] for xi in x]
        z = torch.empty((B, sum(sizes), self.no), dtype=torch.float32, device=x[0].device)
        off = 0
        for i in range(self.nl):
            apx = (self.anchors[i] * self.stride[i]).float().contiguous()   # anchor_grid values (:133)
            ops.detect_decode(x[i], apx, float(self.stride[i]), z, off)
            off += sizes[i]
        return (z, x)
