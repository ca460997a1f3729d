"""SSOD detector (host-side mirror of reference models/detector/yolo_ssod.py:44-118, 158-238):
the YoloV5 detector plus three ``netD`` domain heads behind a gradient reversal; ``forward`` returns
``(out, [netD(P3), netD(P4), netD(P5)])`` with each feature (B, 2, H, W)."""
import torch
import torch.nn as nn

from ... import ops
from ...autograd import ConvBiasFn, GradReverseFn
from .yolo import Model as _BaseModel


def conv1x1(in_planes, out_planes, stride):
    "1x1 convolution with padding"
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=0, bias=False)


class netD(nn.Module):
    def __init__(self, channel, ratio, context=False):
        super(netD, self).__init__()
        self.ratio = ratio
        self.conv1 = conv1x1(int(channel * self.ratio), int(channel * self.ratio), stride=1)
        self.conv2 = conv1x1(int(channel * self.ratio), 2, stride=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        c1, c2 = self.conv1._et_slot, self.conv2._et_slot
        if torch.is_grad_enabled() and x.requires_grad:
            h = ConvBiasFn.apply(x, self.conv1.weight, c1, ops.ACT_RELU, None)
            y = ConvBiasFn.apply(h, self.conv2.weight, c2, ops.ACT_NONE, None)
        else:
            h = ops.conv2d_fwd(x, c1.w_lp, 1, 0, act=ops.ACT_RELU)
            y = ops.conv2d_fwd(h, c2.w_lp, 1, 0)
        return y[..., :2].permute(0, 3, 1, 2)     # (B, 2, H, W) view of the NHWC result


class Model(_BaseModel):
    def _build_extra(self, cfg):
        self.det_8 = netD(cfg.Model.Neck.out_channels[0], cfg.Model.width_multiple)
        self.det_16 = netD(cfg.Model.Neck.out_channels[1], cfg.Model.width_multiple)
        self.det_32 = netD(cfg.Model.Neck.out_channels[2], cfg.Model.width_multiple)
        # SSOD.with_da_loss False (every shipped recipe) multiplies the domain losses by 0
        # (trainer/ssod_trainer.py:633-636): their gradient is exactly zero, so the netD branch is
        # evaluated without an autograd graph; with_da_loss True needs the reversal backward.
        self.da_grad = bool(cfg.SSOD.with_da_loss)

    def _forward_once(self, x, profile=False, visualize=False):
        feats = self._features(x)
        out = self.head(feats)
        f8, f16, f32 = feats
        if self.da_grad and self.training:
            feature = [self.det_8(GradReverseFn.apply(f8)), self.det_16(GradReverseFn.apply(f16)),
                       self.det_32(GradReverseFn.apply(f32))]
        else:
            with torch.no_grad():
                feature = [self.det_8(f8.detach()), self.det_16(f16.detach()), self.det_32(f32.detach())]
        return out, feature
