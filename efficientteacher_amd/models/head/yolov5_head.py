"""Anchor-based Detect head (host-side mirror of reference models/head/yolov5_head.py:7-87,127-136)."""
import math

import torch
import torch.nn as nn

from ... import ops
from ...autograd import ConvBiasFn, head_view


class Detect(nn.Module):
    stride = None  # strides computed during build

    def __init__(self, cfg):  # detection layer
        super(Detect, self).__init__()
        self.nc = cfg.Dataset.nc  # number of classes
        self.num_keypoints = cfg.Dataset.np
        if self.num_keypoints:
            raise NotImplementedError("keypoint heads are outside the hot path")
        self.cur_imgsize = [cfg.Dataset.img_size, cfg.Dataset.img_size]
        anchors = cfg.Model.anchors
        ch = [int(out_c * cfg.Model.width_multiple) for out_c in cfg.Model.Neck.out_channels]
        self.no = self.nc + self.num_keypoints + 5  # number of outputs per anchor
        self.nl = len(anchors)  # number of detection layers
        self.na = len(anchors[0]) // 2  # number of anchors
        self.grid = [torch.zeros(1)] * self.nl
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))  # shape(nl,na,2)
        self.anchor_grid = [torch.zeros(1)] * self.nl
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)  # output conv
        self.stride = cfg.Model.Head.strides
        self.export = False

    def initialize_biases(self, cf=None):  # initialize biases into Detect(), cf is class frequency
        # https://arxiv.org/abs/1708.02002 section 3.3
        for mi, s in zip(self.m, self.stride):
            b = mi.bias.view(self.na, -1)  # conv.bias(255) to (3,85)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)  # obj (8 objects per 640 image)
            b.data[:, 5:] += math.log(0.6 / (self.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())  # cls
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

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
