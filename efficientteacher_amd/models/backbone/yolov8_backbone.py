"""YOLOv8 backbone (host-side mirror of reference models/backbone/yolov8_backbone.py:25-100): the YOLOv5 stem and
stride-2 convs with C2f blocks; widths make_divisible(c * width_multiple, 8), depths max(round(n * depth_multiple), 1).
Module names (and hence state_dict keys) are the reference's."""
import torch.nn as nn

from ...utils.general import make_divisible
from .common import C2f, Conv, SPPF


class YoloV8BackBone(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.gd = cfg.Model.depth_multiple
        self.gw = cfg.Model.width_multiple
        self.channels_out = {'stage1': 64, 'stage2_1': 128, 'stage2_2': 128, 'stage3_1': 256, 'stage3_2': 256,
                             'stage4_1': 512, 'stage4_2': 512, 'stage5': 768, 'spp': 768, 'csp1': 768}
        self.re_channels_out()
        act = {'SiLU': 'silu', 'ReLU': 'relu'}.get(cfg.Model.Backbone.activation)
        if act is None:
            raise NotImplementedError("hard_swish has no gfx950 kernel (SiLU / ReLU only)")
        c = self.channels_out
        self.stage1 = Conv(3, c['stage1'], 6, 2, 2, 1, act)
        self.stage2_1 = Conv(c['stage1'], c['stage2_1'], 3, 2, None, 1, act)
        self.stage2_2 = C2f(c['stage2_1'], c['stage2_2'], self.get_depth(3), True, 1, 0.5, act)
        self.stage3_1 = Conv(c['stage2_2'], c['stage3_1'], 3, 2, None, 1, act)
        self.stage3_2 = C2f(c['stage3_1'], c['stage3_2'], self.get_depth(6), True, 1, 0.5, act)
        self.stage4_1 = Conv(c['stage3_2'], c['stage4_1'], 3, 2, None, 1, act)
        self.stage4_2 = C2f(c['stage4_1'], c['stage4_2'], self.get_depth(6), True, 1, 0.5, act)
        self.stage5_1 = Conv(c['stage4_2'], c['stage5'], 3, 2, None, 1, act)
        self.stage5_2 = C2f(c['stage5'], c['csp1'], self.get_depth(3), True, 1, 0.5, act)
        self.sppf = SPPF(c['csp1'], c['spp'], 5, act)
        self.out_shape = {'C3_size': c['stage3_2'], 'C4_size': c['stage4_2'], 'C5_size': c['spp']}

    def forward(self, x):
        x22 = self.stage2_2(self.stage2_1(self.stage1(x)))
        c3 = self.stage3_2(self.stage3_1(x22))
        c4 = self.stage4_2(self.stage4_1(c3))
        return c3, c4, self.sppf(self.stage5_2(self.stage5_1(c4)))

    def get_depth(self, n):
        return max(round(n * self.gd), 1) if n > 1 else n

    def get_width(self, n):
        return make_divisible(n * self.gw, 8)

    def re_channels_out(self):
        for k, v in self.channels_out.items():
            self.channels_out[k] = self.get_width(v)
