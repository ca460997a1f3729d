"""what do the buffer-form mismatches of conv_gemm_rs_kernel<128, 64> look like, per workgroup tile?  For every launch that differs from the flat form:
per 128-pixel tile the set of wrong pixels (offset inside the tile) and channels, reduced to a pattern key and counted."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pat = collections.Counter()
launches = failing = 0
for (B, h, cin, cout) in [(8, 160, 64, 64), (8, 160, 128, 64)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, h, h, cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((cout, 3, 3, cin), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    os.environ["ET_CONV_BUF_DMA"] = "0"
    ref = ops.conv2d_fwd(x, w, 1, 1)
    os.environ["ET_CONV_BUF_DMA"] = "1"
    for rep in range(reps):
        y = ops.conv2d_fwd(x, w, 1, 1)
        launches += 1
        bad = (y != ref).view(-1, cout)
        if not bool(bad.any()):
            continue
        failing += 1
        rows = bad.any(1).nonzero().flatten()
        for t in torch.unique(rows // 128).tolist():
            sub = bad[t * 128:(t + 1) * 128]
            pr = sub.any(1).nonzero().flatten().tolist()
            ch = sub.any(0).nonzero().flatten().tolist()
            d = (y.view(-1, cout)[t * 128:(t + 1) * 128].float() - ref.view(-1, cout)[t * 128:(t + 1) * 128].float()).abs().max().item()
            pat[(len(pr), pr[0], pr[-1], len(ch), ch[0], ch[-1], "big" if d > 0.25 else "small")] += 1
print("launches", launches, "failing", failing)
print("pattern: (wrong pixels in the tile, first, last, wrong channels, first, last, max |diff| > 0.25?) x count")
for k, v in pat.most_common(40):
    print("  ", k, "x", v)
