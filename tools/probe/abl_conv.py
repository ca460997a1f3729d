"""Time a few conv layers with whatever library ET_HIP_LIB points at (ablation builds: tools/probe/build_ablate.sh).
usage: ET_HIP_LIB=... python tools/probe/abl_conv.py "cin,cout,k,s,h,B;..." """
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops  # noqa: E402

shapes = [tuple(int(v) for v in t.split(",")) for t in (sys.argv[1] if len(sys.argv) > 1 else "256,256,3,1,40,64;1024,1024,1,1,20,64").split(";")]
dev = torch.device("cuda:0")
out = {}
for (cin, cout, k, s, h, B) in shapes:
    p = k // 2
    x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(torch.bfloat16)
    oh, ow = ops.conv_out_hw(h, h, k, s, p)
    y = torch.empty(B, oh, ow, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv2d_fwd(x, w, s, p, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(10):
            ops.conv2d_fwd(x, w, s, p, out=y)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * oh * ow * cout * cin * k * k
    out[f"{cin}x{cout}k{k}s{s}h{h}"] = dict(us=round(best * 1e3, 1), tf=round(fl / best / 1e9, 1), kern=ops.kernel_name("fwd", torch.bfloat16, B, h, h, cin, cout, k, s, p))
print("ABL", os.environ.get("ET_HIP_LIB", "default"), os.environ.get("ET_CONV_PP", ""), json.dumps(out))
