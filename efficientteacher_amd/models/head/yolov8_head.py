"""YOLOv8 decoupled anchor-free head (host-side mirror of reference models/head/yolov8_head.py:10-220).

train : (feats, cls (B, A, nc) logits, reg (B, A, 4*(reg_max+1)) DFL logits), A = sum of H*W over the levels -- the NHWC
        GEMM outputs ARE (B, H*W, C), so the reference's flatten(2).permute(0, 2, 1) costs nothing here.
eval  : (z (B, A, 5+nc) [cx, cy, w, h, 1, sigmoid(cls)], (feats, cls, reg)) -- DFL expectation + dist2bbox + stride in
        et_v8_decode (csrc/tal.hip).
"""
import math

import torch
import torch.nn as nn

from ... import ops
from ...autograd import ConvBiasFn
from ..backbone.common import Conv


class _OutConv(nn.Conv2d):
    """the bias-carrying 1x1 output conv of a branch, run through the implicit-GEMM kernel"""

    def forward(self, x):
        cs = getattr(self, "_et_slot", None)
        if cs is None:
            raise RuntimeError("model state is not on the device arenas yet (model.to('cuda')); no CPU path")
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad) and self.training:
            y = ConvBiasFn.apply(x, self.weight, cs, ops.ACT_NONE, None)
        else:
            y = ops.conv2d_fwd(x, cs.w_lp, 1, 0, bias=cs.bias)
        return y[..., :self.out_channels]


class YoloV8Detect(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        num_anchors = len(cfg.Model.anchors) if isinstance(cfg.Model.anchors, (list, tuple)) else cfg.Model.anchors
        self.nc = cfg.Dataset.nc
        self.no = self.nc + 5
        self.nl = cfg.Model.Neck.num_outs
        self.num_keypoints = cfg.Dataset.np
        self.na = num_anchors
        self.prune = False
        self.use_l1 = False
        self.export = False
        self.prior_prob = 1e-2
        self.inplace = cfg.Model.inplace
        self.reg_max = cfg.Loss.reg_max
        self.use_dfl = cfg.Loss.use_dfl
        self.stride = torch.Tensor(cfg.Model.Head.strides)
        self.proj_conv = nn.Conv2d(self.reg_max + 1, 1, 1, bias=False)
        self.grid_cell_offset = cfg.Loss.grid_cell_offset
        self.grid_cell_size = cfg.Loss.grid_cell_size
        act = {'SiLU': 'silu', 'ReLU': 'relu'}.get(cfg.Model.Head.activation)
        if act is None:
            raise NotImplementedError("hard_swish has no gfx950 kernel (SiLU / ReLU only)")
        ch = [int(out_c * cfg.Model.width_multiple) for out_c in cfg.Model.Neck.out_channels]
        c2, c3 = max((16, ch[0] // 4, (self.reg_max + 1) * 4)), max(ch[0], self.nc)
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3, 1, None, 1, act=act), Conv(c2, c2, 3, 1, None, 1, act=act),
                                               _OutConv(c2, 4 * (self.reg_max + 1), 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3, 1, None, 1, act=act), Conv(c3, c3, 3, 1, None, 1, act=act),
                                               _OutConv(c3, self.nc, 1)) for x in ch)

    def initialize_biases(self):
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0  # box
            b[-1].bias.data[:self.nc] = math.log(5 / self.nc / (640 / s) ** 2)  # cls (.01 objects, 80 classes, 640 img)
        self.proj = nn.Parameter(torch.linspace(0, self.reg_max, self.reg_max + 1), requires_grad=False)
        self.proj_conv.weight = nn.Parameter(self.proj.view([1, self.reg_max + 1, 1, 1]).clone().detach(), requires_grad=False)

    def forward(self, x):
        if self.export:
            raise NotImplementedError("export path is out of scope")
        x = list(x)
        regs, clss = [], []
        for i in range(self.nl):
            regs.append(self.cv2[i](x[i]))           # (B, H, W, 4*(reg_max+1)) NHWC
            clss.append(self.cv3[i](x[i]))           # (B, H, W, nc)
        B = x[0].shape[0]
        cls_score_list = torch.cat([c.reshape(B, -1, c.shape[3]) for c in clss], 1)
        reg_distri_list = torch.cat([r.reshape(B, -1, r.shape[3]) for r in regs], 1)
        feats = [f.permute(0, 3, 1, 2) for f in x]    # the reference hands NCHW feature maps on (their H, W seed the anchors)
        if self.training:
            return feats, cls_score_list, reg_distri_list
        if not self.use_dfl:
            raise NotImplementedError("Loss.use_dfl False: every shipped YOLOv8 recipe uses the DFL head")
        sizes = [r.shape[1] * r.shape[2] for r in regs]
        z = torch.empty((B, sum(sizes), self.no), dtype=torch.float32, device=x[0].device)
        off = 0
        for i in range(self.nl):
            ops.v8_decode(regs[i], clss[i], self.reg_max, self.nc, float(self.stride[i]), self.grid_cell_offset, z, off)
            off += sizes[i]
        return z, (feats, cls_score_list, reg_distri_list)
