"""is the 1x1 stream kernel of two libraries bit-identical?  run with ET_HIP_LIB=..., writes outputs to argv[1]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops
dev = torch.device("cuda:0")
out = {}
for dt in (torch.float16, torch.bfloat16):
    for (B, h, cin, cout) in [(4, 160, 64, 64), (4, 80, 128, 128), (4, 40, 256, 256), (2, 80, 128, 64), (4, 160, 64, 32)]:
        g = torch.Generator().manual_seed(7)
        x = torch.randn((B, h, h, cin), generator=g).to(dt).to(dev)
        w = (torch.randn((cout, 1, 1, cin), generator=g) * cin ** -0.5).to(dt).to(dev)
        name = ops.kernel_name("fwd", dt, B, h, h, cin, cout, 1, 1, 0)
        y, st = ops.conv2d_fwd(x, w, 1, 0, want_stats=True)
        out[f"{dt}-{B}-{h}-{cin}-{cout}-y"] = y.cpu(); out[f"{dt}-{B}-{h}-{cin}-{cout}-st"] = st.cpu()
        dy = torch.randn((B, h, h, cout), generator=g).to(dt).to(dev)
        wT = ops.weight_transpose(w)
        out[f"{dt}-{B}-{h}-{cin}-{cout}-dx"] = ops.conv2d_dgrad(dy, wT, (h, h), 1, 0).cpu()
        dw = torch.zeros((cout, 1, 1, cin), dtype=torch.float32, device=dev)
        ops.conv2d_wgrad(x, dy, dw, 1, 1, 0)
        torch.cuda.synchronize()
        out[f"{dt}-{B}-{h}-{cin}-{cout}-dw"] = dw.cpu()
        print(name, flush=True)
torch.save(out, sys.argv[1])
