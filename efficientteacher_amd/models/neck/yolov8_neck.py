"""YOLOv8 PAN neck (host-side mirror of reference models/neck/yolov8_neck.py:6-118): no lateral 1x1 convs, C2f blocks."""
import torch.nn as nn

from ...autograd import UpsampleCatFn
from ...utils.general import make_divisible
from ..backbone.common import C2f, Concat, Conv


class YoloV8Neck(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.gd = cfg.Model.depth_multiple
        self.gw = cfg.Model.width_multiple
        input_p3, input_p4, input_p5 = cfg.Model.Neck.in_channels
        output_p3, output_p4, output_p5 = cfg.Model.Neck.out_channels
        self.channels = {'input_p3': input_p3, 'input_p4': input_p4, 'input_p5': input_p5,
                         'output_p3': output_p3, 'output_p4': output_p4, 'output_p5': output_p5}
        self.re_channels_out()
        c = self.channels
        self.input_p3, self.input_p4, self.input_p5 = c['input_p3'], c['input_p4'], c['input_p5']
        self.output_p3, self.output_p4, self.output_p5 = c['output_p3'], c['output_p4'], c['output_p5']
        act = {'SiLU': 'silu', 'ReLU': 'relu'}.get(cfg.Model.Neck.activation)
        if act is None:
            raise NotImplementedError("hard_swish has no gfx950 kernel (SiLU / ReLU only)")
        self.upsample1 = nn.Upsample(scale_factor=2, mode="nearest")
        self.C1 = C2f(self.input_p5 + self.input_p4, self.input_p4, self.get_depth(3), False, 1, 0.5, act)
        self.upsample2 = nn.Upsample(scale_factor=2, mode="nearest")
        self.C2 = C2f(self.input_p4 + self.input_p3, self.output_p3, self.get_depth(3), False, 1, 0.5, act)
        self.conv3 = Conv(self.output_p3, self.output_p3, 3, 2, None, 1, act)
        self.C3 = C2f(self.output_p3 + self.input_p4, self.output_p4, self.get_depth(3), False, 1, 0.5, act)
        self.conv4 = Conv(self.output_p4, self.output_p4, 3, 2, None, 1, act)
        self.C4 = C2f(self.output_p4 + self.input_p5, self.output_p5, self.get_depth(3), False, 1, 0.5, act)
        self.concat = Concat()

    def get_depth(self, n):
        return max(round(n * self.gd), 1) if n > 1 else n

    def get_width(self, n):
        return make_divisible(n * self.gw, 8)

    def re_channels_out(self):
        for k, v in self.channels.items():
            self.channels[k] = self.get_width(v)

    def forward(self, inputs):
        P3, P4, P5 = inputs
        x1 = self.C1(UpsampleCatFn.apply(P5, P4))           # upsample1 + concat written straight into one buffer
        x2 = self.C2(UpsampleCatFn.apply(x1, P3))
        x3 = self.C3(self.concat([self.conv3(x2), x1]))
        x4 = self.C4(self.concat([self.conv4(x3), P5]))
        return [x2, x3, x4]
