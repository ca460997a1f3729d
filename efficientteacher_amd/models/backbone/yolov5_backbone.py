"""CSPDarknet backbone (host-side mirror of reference models/backbone/yolov5_backbone.py:26-98)."""
import torch.nn as nn

from ...autograd import GradFork

from ...utils.general import make_divisible
from .common import C3, SPPF, Conv


STAGE_EVENTS = None      # dict filled with a HIP event per pyramid stage of the NEXT training forward (trainer: teacher start point)


def _mark_stage(name, t):
    if STAGE_EVENTS is not None and t.is_cuda and t.requires_grad:
        import torch
        e = torch.cuda.Event()
        e.record()
        STAGE_EVENTS[name] = e


class YoloV5BackBone(nn.Module):
    supports_concat_dst = True       # forward(dst_c3=, dst_c4=): C3 / C4 written straight into the neck's concat buffers (detector/yolo.py)

    def __init__(self, cfg):
        super(YoloV5BackBone, self).__init__()
        self.gd = cfg.Model.depth_multiple
        self.gw = cfg.Model.width_multiple
        self.channels_out = {'stage1': 64, 'stage2_1': 128, 'stage2_2': 128, 'stage3_1': 256, 'stage3_2': 256,
                             'stage4_1': 512, 'stage4_2': 512, 'stage5': 1024, 'spp': 1024, 'csp1': 1024,
                             'conv1': 1024}
        self.re_channels_out()
        if cfg.Model.Backbone.activation == 'SiLU':
            CONV_ACT, C_ACT = 'silu', 'silu'
        elif cfg.Model.Backbone.activation == 'ReLU':
            CONV_ACT, C_ACT = 'relu', 'relu'
        else:
            CONV_ACT, C_ACT = 'hard_swish', 'relu_hswish'
        c = self.channels_out
        self.stage1 = Conv(3, c['stage1'], 6, 2, 2, 1, CONV_ACT)
        self.stage2_1 = Conv(c['stage1'], c['stage2_1'], 3, 2, None, 1, CONV_ACT)
        self.stage2_2 = C3(c['stage2_1'], c['stage2_2'], self.get_depth(3), True, 1, 0.5, C_ACT)
        self.stage3_1 = Conv(c['stage2_2'], c['stage3_1'], 3, 2, None, 1, CONV_ACT)
        self.stage3_2 = C3(c['stage3_1'], c['stage3_2'], self.get_depth(6), True, 1, 0.5, C_ACT)
        self.stage4_1 = Conv(c['stage3_2'], c['stage4_1'], 3, 2, None, 1, CONV_ACT)
        self.stage4_2 = C3(c['stage4_1'], c['stage4_2'], self.get_depth(9), True, 1, 0.5, C_ACT)
        self.stage5_1 = Conv(c['stage4_2'], c['stage5'], 3, 2, None, 1, CONV_ACT)
        self.stage5_2 = C3(c['stage5'], c['csp1'], self.get_depth(3), True, 1, 0.5, C_ACT)
        self.sppf = SPPF(c['csp1'], c['spp'], 5, CONV_ACT)
        self.out_shape = {'C3_size': c['stage3_2'], 'C4_size': c['stage4_2'], 'C5_size': c['conv1']}

    def forward(self, x, dst_c3=None, dst_c4=None):
        """dst_c3 / dst_c4 = (buffer, channel offset): produce C3 / C4 in place inside the neck's concat buffers"""
        x1 = self.stage1(x)        # P1/2
        _mark_stage("p1", x1)
        x21 = self.stage2_1(x1)    # P2/4
        x22 = self.stage2_2(x21)
        _mark_stage("p2", x22)
        x31 = self.stage3_1(x22)   # P3/8
        # C3 / C4 feed the next stage AND the neck: the stride-2 conv adds its input gradient into the neck's (autograd.GradFork)
        c3, c3n, f3 = GradFork.split(self.stage3_2(x31, dst=dst_c3))
        _mark_stage("p3", c3)
        x41 = self.stage4_1(c3, acc=f3)    # P4/16
        c4, c4n, f4 = GradFork.split(self.stage4_2(x41, dst=dst_c4))
        _mark_stage("p4", c4)
        x51 = self.stage5_1(c4, acc=f4)    # P5/32
        x5 = self.stage5_2(x51)
        return GradFork.tap(c3n, f3), GradFork.tap(c4n, f4), self.sppf(x5)

    def get_depth(self, n):
        return max(round(n * self.gd), 1) if n > 1 else n

    def get_width(self, n):
        return make_divisible(n * self.gw, 8)

    def re_channels_out(self):
        for k, v in self.channels_out.items():
            self.channels_out[k] = self.get_width(v)
