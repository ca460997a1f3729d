"""Is the BatchNorm + SiLU elementwise family HBM-bound or VALU-bound?  The same passes with ACT_SILU and ACT_NONE on COLD tensors
(four rotating 210 MB buffers: 64 x 80 x 80 x 128 bf16), HIP events."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops
dev = torch.device("cuda:0")
B, H, C = 64, 80, 128
R = 6
ys = [torch.randn(B, H, H, C, device=dev).to(torch.bfloat16) for _ in range(R)]
dzs = [torch.randn(B, H, H, C, device=dev).to(torch.bfloat16) for _ in range(R)]
outs = [torch.empty_like(ys[0]) for _ in range(R)]
gamma = torch.rand(C, device=dev) + 0.5
mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev)
scale = gamma * invstd; shift = -mean * scale
dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
nb = ys[0].numel() * 2
res = {}
def timeit(fn, n=3 * R):
    for i in range(R): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % R)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for act, name in ((ops.ACT_SILU, "silu"), (ops.ACT_NONE, "none")):
    tf = timeit(lambda i: ops.bn_act_fwd(ys[i], scale, shift, act, out=outs[i]))
    tb = timeit(lambda i: ops.bn_act_bwd(dzs[i], ys[i], gamma, scale, shift, mean, invstd, act, dg, db, out=outs[i]))
    part = torch.zeros((1600, 2, C), device=dev)
    ta = timeit(lambda i: ops.bn_act_bwd(dzs[i], ys[i], gamma, scale, shift, mean, invstd, act, dg, db, out=outs[i], partial=part))
    res[name] = dict(fwd_us=tf * 1e6, fwd_TBps=2 * nb / tf / 1e12, bwd_us=tb * 1e6, bwd_TBps=5 * nb / tb / 1e12,
                     finalize_apply_us=ta * 1e6, apply_TBps=3 * nb / ta / 1e12, reduce_us=(tb - ta) * 1e6, reduce_TBps=2 * nb / (tb - ta) / 1e12)
    print(name, res[name], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bn_act_cost.json", "w"), indent=1)
