"""SPPF's pool chain at the step's size: 64 x 20 x 20 x 512 (bf16), three forward + three backward launches, us per launch"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops
dev = torch.device("cuda:0")
for B in (64, 32):
    C = 512
    cat = torch.randn((B, 20, 20, 4 * C), device=dev).to(torch.bfloat16)
    g = torch.randn((B, 20, 20, 4 * C), device=dev).to(torch.bfloat16)
    def chain():
        idx = []
        for i in range(3):
            _, ix = ops.maxpool5_fwd(cat[..., i * C:(i + 1) * C], out=cat[..., (i + 1) * C:(i + 2) * C]); idx.append(ix)
        return idx
    idx = chain()
    def back():
        d2 = ops.maxpool5_bwd(g[..., 3 * C:], idx[2], base=g[..., 2 * C:3 * C])
        d1 = ops.maxpool5_bwd(d2, idx[1], base=g[..., C:2 * C])
        return ops.maxpool5_bwd(d1, idx[0], base=g[..., :C])
    for name, fn in (("fwd", chain), ("bwd", back)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1) / 150 * 1e3:7.1f} us per launch", flush=True)
