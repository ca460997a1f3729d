"""One conv shape launched a few times, for rocprofv3 --pmc passes (HBM traffic of the dominant kernel).
usage: python tools/pmc_conv.py [cin cout k s h B]   (default: 256->256 3x3 s1 @40^2, B=64: the largest
YOLOv5l bucket)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientteacher_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:7]] + [256, 256, 3, 1, 40, 64][len(sys.argv) - 1:]
cin, cout, k, s, h, B = a
dev = torch.device("cuda:0")
x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(torch.bfloat16)
p = 2 if k == 6 else k // 2
oh, ow = ops.conv_out_hw(h, h, k, s, p)
y = torch.empty(B, oh, ow, cout, device=dev, dtype=torch.bfloat16)
dy = torch.randn(B, oh, ow, cout, device=dev).to(torch.bfloat16)
dw = torch.zeros(cout, k, k, cin, device=dev)
wT = ops.weight_transpose(w)
dx = torch.empty_like(x)
for _ in range(5):
    ops.conv2d_fwd(x, w, s, p, out=y)
    ops.conv2d_dgrad(dy, wT, (h, h), s, p, out=dx)
    ops.conv2d_wgrad(x, dy, dw, k, s, p)
torch.cuda.synchronize()
flop = 2.0 * B * oh * ow * cout * cin * k * k
print(f"shape cin={cin} cout={cout} k={k} s={s} h={h} B={B}: {flop/1e9:.1f} GFLOP/launch; algorithmic bytes fwd "
      f"{(x.numel() + y.numel() + w.numel()) * 2 / 1e6:.1f} MB")
