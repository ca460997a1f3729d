"""TEST INFRASTRUCTURE -- plain-PyTorch (CPU, fp32) restatement of the SSOD detector.

Restates models/detector/yolo_ssod.py:44-118 (Model), models/backbone/common.py:471-708 (Conv,
Bottleneck, C3, SPPF), models/backbone/yolov5_backbone.py:26-98, models/neck/yolov5_neck.py:6-109,
models/head/yolov5_head.py:7-87 and the netD heads (yolo_ssod.py:224-238) with stock torch.nn ops,
with the reference's module names so that a state_dict moves freely between the reference, this
restatement and the HIP-backed model.  Used as the oracle of model-level tests at sizes the golden
files do not cover and as the ``cpu_baseline`` ("port") of bench.py.  Pinned against
tests/golden/model_tiny.npz in tests/test_oracle_golden.py.
"""
import math

import torch
import torch.nn as nn

from . import detect as o_det


def make_divisible(x, d):
    return math.ceil(x / d) * d


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, p=None):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C3(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, e=1.0) for _ in range(n)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat([x, y1, y2, self.m(y2)], 1))


class Backbone(nn.Module):
    def __init__(self, gw, gd):
        super().__init__()
        w = lambda c: make_divisible(c * gw, 8)
        d = lambda n: max(round(n * gd), 1)
        self.stage1 = Conv(3, w(64), 6, 2, 2)
        self.stage2_1 = Conv(w(64), w(128), 3, 2)
        self.stage2_2 = C3(w(128), w(128), d(3))
        self.stage3_1 = Conv(w(128), w(256), 3, 2)
        self.stage3_2 = C3(w(256), w(256), d(6))
        self.stage4_1 = Conv(w(256), w(512), 3, 2)
        self.stage4_2 = C3(w(512), w(512), d(9))
        self.stage5_1 = Conv(w(512), w(1024), 3, 2)
        self.stage5_2 = C3(w(1024), w(1024), d(3))
        self.sppf = SPPF(w(1024), w(1024), 5)

    def forward(self, x):
        x = self.stage2_2(self.stage2_1(self.stage1(x)))
        c3 = self.stage3_2(self.stage3_1(x))
        c4 = self.stage4_2(self.stage4_1(c3))
        return c3, c4, self.sppf(self.stage5_2(self.stage5_1(c4)))


class Neck(nn.Module):
    def __init__(self, gw, gd):
        super().__init__()
        w = lambda c: make_divisible(c * gw, 8)
        d = lambda n: max(round(n * gd), 1)
        p3, p4, p5 = w(256), w(512), w(1024)
        self.conv1 = Conv(p5, p5 // 2, 1, 1)
        self.upsample1 = nn.Upsample(scale_factor=2, mode="nearest")
        self.C1 = C3(p5 // 2 + p4, p4, d(3), False)
        self.conv2 = Conv(p4, p3, 1, 1)
        self.upsample2 = nn.Upsample(scale_factor=2, mode="nearest")
        self.C2 = C3(p3 + p3, p3, d(3), False)
        self.conv3 = Conv(p3, p3, 3, 2)
        self.C3 = C3(p3 + p3, p4, d(3), False)
        self.conv4 = Conv(p4, p4, 3, 2)
        self.C4 = C3(p4 + p5 // 2, p5, d(3), False)

    def forward(self, inputs):
        P3, P4, P5 = inputs
        xp_1 = self.conv1(P5)
        x1 = self.C1(torch.cat([self.upsample1(xp_1), P4], 1))
        xp_2 = self.conv2(x1)
        x2 = self.C2(torch.cat([self.upsample2(xp_2), P3], 1))
        x3 = self.C3(torch.cat([self.conv3(x2), xp_2], 1))
        x4 = self.C4(torch.cat([self.conv4(x3), xp_1], 1))
        return x2, x3, x4


class Detect(nn.Module):
    def __init__(self, nc, anchors, ch):
        super().__init__()
        self.nc, self.no, self.nl, self.na = nc, nc + 5, len(anchors), len(anchors[0]) // 2
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.stride = torch.tensor([8., 16., 32.])

    def forward(self, x):
        xs = [o_det.permute_raw(self.m[i](x[i]), self.na, self.no) for i in range(self.nl)]
        if self.training:
            return xs
        return o_det.decode([t.clone() for t in xs], self.anchors, self.stride), xs


class NetD(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 1, bias=False)
        self.conv2 = nn.Conv2d(c, 2, 1, bias=False)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class Model(nn.Module):
    """width/depth multiples + anchors (pixels) + nc; ``from_cfg`` reads them from a CfgNode."""

    def __init__(self, gw=1.0, gd=1.0, nc=80, anchors=None):
        super().__init__()
        anchors = anchors or [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
        self.backbone = Backbone(gw, gd)
        self.neck = Neck(gw, gd)
        ch = [int(c * gw) for c in (256, 512, 1024)]
        self.head = Detect(nc, anchors, ch)
        self.det_8, self.det_16, self.det_32 = NetD(ch[0]), NetD(ch[1]), NetD(ch[2])
        self.head.anchors /= self.head.stride.view(-1, 1, 1)
        self.stride = self.head.stride
        for mi, s in zip(self.head.m, self.head.stride):       # initialize_biases, yolov5_head.py:36-45
            b = mi.bias.view(self.head.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (nc - 0.99))

    @classmethod
    def from_cfg(cls, cfg):
        return cls(cfg.Model.width_multiple, cfg.Model.depth_multiple, cfg.Dataset.nc, cfg.Model.anchors)

    def forward(self, x):
        feats = self.neck(self.backbone(x))
        out = self.head(feats)
        feature = [self.det_8(feats[0]), self.det_16(feats[1]), self.det_32(feats[2])]
        return out, feature
