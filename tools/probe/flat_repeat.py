"""Assurance run for the shipped (flat-address) LDS-DMA kernels after the buffer-form finding: the forward of each tile family N times on the
same operands -- the launches have no atomics, so every output must be bit-equal to the first one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for (B, h, cin, cout, k, s) in [(64, 160, 64, 64, 3, 1), (4, 320, 64, 64, 3, 1), (64, 80, 128, 128, 3, 1), (64, 40, 256, 256, 3, 1), (64, 20, 512, 512, 3, 1),
                                (64, 20, 1024, 1024, 1, 1), (64, 80, 256, 512, 3, 2), (64, 320, 64, 128, 3, 2), (64, 80, 128, 128, 1, 1), (64, 160, 128, 64, 1, 1),
                                (64, 80, 512, 128, 1, 1)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, h, h, cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((cout, k, k, cin), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    p = k // 2
    ref = ops.conv2d_fwd(x, w, s, p)
    wT = ops.weight_transpose(w)
    oh = ref.shape[1]
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16).to(dev)
    dref = ops.conv2d_dgrad(dy, wT, (h, h), s, p)
    bad = dbad = 0
    for i in range(N):
        bad += int(not torch.equal(ops.conv2d_fwd(x, w, s, p), ref))
        dbad += int(not torch.equal(ops.conv2d_dgrad(dy, wT, (h, h), s, p), dref))
    print(f"{ops.kernel_name('fwd', torch.bfloat16, B, h, h, cin, cout, k, s, p)[:60]:60s} {(B, h, cin, cout, k, s)}  fwd differing {bad} / {N}   dgrad differing {dbad} / {N}", flush=True)
