"""PANet neck (host-side mirror of reference models/neck/yolov5_neck.py:6-109)."""
import torch
import torch.nn as nn

from ...autograd import GradFork, JoinSlicesFn, UpsampleCatFn
from ...utils.general import make_divisible
from ..backbone.common import C3, Concat, Conv


class YoloV5Neck(nn.Module):
    def __init__(self, cfg):
        super(YoloV5Neck, self).__init__()
        self.gd = cfg.Model.depth_multiple
        self.gw = cfg.Model.width_multiple
        input_p3, input_p4, input_p5 = cfg.Model.Neck.in_channels
        output_p3, output_p4, output_p5 = cfg.Model.Neck.out_channels
        self.channels = {'input_p3': input_p3, 'input_p4': input_p4, 'input_p5': input_p5,
                         'output_p3': output_p3, 'output_p4': output_p4, 'output_p5': output_p5}
        self.re_channels_out()
        for k, v in self.channels.items():
            setattr(self, k, v)
        if cfg.Model.Neck.activation == 'SiLU':
            CONV_ACT, C_ACT = 'silu', 'silu'
        elif cfg.Model.Neck.activation == 'ReLU':
            CONV_ACT, C_ACT = 'relu', 'relu'
        else:
            CONV_ACT, C_ACT = 'hard_swish', 'relu_hswish'
        self.conv1 = Conv(self.input_p5, int(self.input_p5 / 2), 1, 1, None, 1, CONV_ACT)
        self.upsample1 = nn.Upsample(scale_factor=2, mode="nearest")
        self.C1 = C3(int(self.input_p5 / 2) + self.input_p4, self.input_p4, self.get_depth(3), False, 1, 0.5, C_ACT)
        self.conv2 = Conv(self.input_p4, self.input_p3, 1, 1, None, 1, CONV_ACT)
        self.upsample2 = nn.Upsample(scale_factor=2, mode="nearest")
        self.C2 = C3(self.input_p3 + self.input_p3, self.output_p3, self.get_depth(3), False, 1, 0.5, C_ACT)
        self.conv3 = Conv(self.output_p3, self.output_p3, 3, 2, None, 1, CONV_ACT)
        self.C3 = C3(self.output_p3 + self.input_p3, self.output_p4, self.get_depth(3), False, 1, 0.5, C_ACT)
        self.conv4 = Conv(self.output_p4, self.output_p4, 3, 2, None, 1, CONV_ACT)
        self.C4 = C3(self.output_p4 + int(self.input_p5 / 2), self.output_p5, self.get_depth(3), False, 1, 0.5, C_ACT)
        self.concat = Concat()

    def get_depth(self, n):
        return max(round(n * self.gd), 1) if n > 1 else n

    def get_width(self, n):
        return make_divisible(n * self.gw, 8)

    def re_channels_out(self):
        for k, v in self.channels.items():
            self.channels[k] = self.get_width(v)

    def concat_slots(self, N, H3, W3, dtype, device):
        """The two top-down concat buffers [upsample(conv1(P5)) | P4] and [upsample(conv2(.)) | P3] allocated BEFORE the backbone
        runs, so that its C4 / C3 blocks write P4 / P3 straight into their halves: ((buf, offset) for C3, (buf, offset) for C4),
        or None when a channel count is not a multiple of 8 (16-byte vectors)."""
        c1o, c2o = self.conv1.conv.out_channels, self.conv2.conv.out_channels
        if c1o % 8 or c2o % 8 or self.input_p3 % 8 or self.input_p4 % 8:
            return None
        cat2 = torch.empty((N, H3, W3, c2o + self.input_p3), dtype=dtype, device=device)
        cat1 = torch.empty((N, H3 // 2, W3 // 2, c1o + self.input_p4), dtype=dtype, device=device)
        return (cat2, c2o), (cat1, c1o)

    def forward(self, inputs, cat_bufs=None):
        """cat_bufs = (buffer holding P3, buffer holding P4) from concat_slots() when the backbone produced them in place"""
        P3, P4, P5 = inputs
        b3, b4 = cat_bufs if cat_bufs is not None else (None, None)
        # the two bottom-up concats [conv3(x2) | xp_2] and [conv4(x3) | xp_1] are produced IN PLACE: the lateral
        # convs write xp_1 / xp_2 into the second half of the concat buffer when they run, the stride-2 convs
        # later fill the first half (JoinSlicesFn is the differentiable "cat" of the filled buffer)
        N, H5, W5, _ = P5.shape
        c1o = self.conv1.conv.out_channels
        if c1o % 8 or self.conv4.conv.out_channels % 8 or self.conv2.conv.out_channels % 8 or self.conv3.conv.out_channels % 8:
            xp_1 = self.conv1(P5)
            x1 = self.C1(UpsampleCatFn.apply(xp_1, P4, None, b4))
            xp_2 = self.conv2(x1)
            x2 = self.C2(UpsampleCatFn.apply(xp_2, P3, None, b3))
            x3 = self.C3(self.concat([self.conv3(x2), xp_2]))
            x4 = self.C4(self.concat([self.conv4(x3), xp_1]))
            return x2, x3, x4
        c4o = self.conv4.conv.out_channels
        buf4 = torch.empty((N, H5, W5, c4o + c1o), dtype=P5.dtype, device=P5.device)
        # xp_1 / xp_2 (lateral outputs) and x2 / x3 (pyramid outputs) have two consumers each; the one that runs its backward LAST
        # (upsample-concat, stride-2 conv) adds its gradient into the other's in place (autograd.GradFork) instead of a torch add
        xp_1u, xp_1, f1 = GradFork.split(self.conv1(P5, dst=(buf4, c4o)))
        x1 = self.C1(UpsampleCatFn.apply(xp_1u, P4, f1, b4))     # upsample1 + concat, no intermediate tensor
        xp_1 = GradFork.tap(xp_1, f1)
        c2o, c3o = self.conv2.conv.out_channels, self.conv3.conv.out_channels
        buf3 = torch.empty((N, x1.shape[1], x1.shape[2], c3o + c2o), dtype=x1.dtype, device=x1.device)
        xp_2u, xp_2, f2 = GradFork.split(self.conv2(x1, dst=(buf3, c3o)))
        x2c, x2, g2 = GradFork.split(self.C2(UpsampleCatFn.apply(xp_2u, P3, f2, b3)))     # upsample2 + concat
        xp_2 = GradFork.tap(xp_2, f2)
        t3 = self.conv3(x2c, dst=(buf3, 0), acc=g2)
        x2 = GradFork.tap(x2, g2)
        x3c, x3, g3 = GradFork.split(self.C3(self._join(buf3, t3, xp_2)))
        t4 = self.conv4(x3c, dst=(buf4, 0), acc=g3)
        x3 = GradFork.tap(x3, g3)
        x4 = self.C4(self._join(buf4, t4, xp_1))
        return x2, x3, x4

    @staticmethod
    def _join(buf, a, b):
        if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
            return JoinSlicesFn.apply((buf,), a, b)
        return buf
