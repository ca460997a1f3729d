"""where do the buffer-descriptor and flat-address forms of conv_gemm_rs_kernel differ?  (same inputs, ET_CONV_BUF_DMA toggled per call)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (B, h, cin, cout) in [tuple(int(v) for v in a.split(',')) for a in (sys.argv[1:] or ['64,160,64,64', '8,160,64,64', '64,80,128,128', '64,40,64,64'])]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, h, h, cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((cout, 3, 3, cin), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    print(ops.kernel_name("fwd", torch.bfloat16, B, h, h, cin, cout, 3, 1, 1), (B, h, cin, cout))
    os.environ["ET_CONV_BUF_DMA"] = "0"
    ref = ops.conv2d_fwd(x, w, 1, 1).float()
    for rep in range(4):
        os.environ["ET_CONV_BUF_DMA"] = "1"
        y = ops.conv2d_fwd(x, w, 1, 1).float()
        torch.cuda.synchronize()
        bad = (y != ref)
        nb = int(bad.sum())
        if nb == 0:
            print("   rep", rep, "identical")
            continue
        idx = bad.nonzero()
        pix = idx[:, 0] * h * h + idx[:, 1] * h + idx[:, 2]
        up = torch.unique(pix)
        print("   rep", rep, "mismatching elements", nb, "pixels", int(up.numel()), "images", torch.unique(idx[:, 0]).tolist()[:10],
              "rows(y)", torch.unique(idx[:, 1]).tolist()[:12], "cols(x) min/max", int(idx[:, 2].min()), int(idx[:, 2].max()),
              "channels", int(torch.unique(idx[:, 3]).numel()), "first pixels", up[:6].tolist(), "tile of first", int(up[0]) // 128,
              "max abs diff", float((y - ref).abs().max()))
