"""GPU micro-benchmarks of the hot-path kernels (run on the MI355X box via gpurun).

  python tools/microbench.py conv   -> per-layer fwd/dgrad/wgrad TFLOP/s for the YOLOv5l conv shapes
                                        at batch B, next to torch's own conv
                                       (MIOpen) on the same data as the known-good reference
  python tools/microbench.py nms    -> non_max_suppression_ssod at (32, 25200, 85)
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientteacher_amd import ops  # noqa: E402

# (Cin, Cout, k, s, Hin, count) -- YOLOv5l SSOD model, appendix A
V5L = [
    (8, 64, 6, 2, 640, 1), (64, 128, 3, 2, 320, 1), (64, 64, 1, 1, 160, 3), (128, 64, 1, 1, 160, 2),
    (128, 128, 1, 1, 160, 1), (64, 64, 3, 1, 160, 3), (128, 256, 3, 2, 160, 1), (128, 128, 1, 1, 80, 9),
    (256, 128, 1, 1, 80, 2), (512, 128, 1, 1, 80, 2), (256, 256, 1, 1, 80, 3), (256, 256, 1, 1, 80, 1),
    (128, 128, 3, 1, 80, 9), (256, 256, 3, 2, 80, 1), (256, 512, 3, 2, 80, 1), (256, 256, 1, 1, 40, 15),
    (512, 256, 1, 1, 40, 5), (512, 512, 1, 1, 40, 4), (1024, 256, 1, 1, 40, 2), (512, 256, 1, 1, 40, 1),
    (256, 256, 3, 1, 40, 15), (512, 512, 3, 2, 40, 1), (512, 1024, 3, 2, 40, 1), (512, 512, 1, 1, 20, 6),
    (1024, 512, 1, 1, 20, 6), (1024, 1024, 1, 1, 20, 3), (2048, 1024, 1, 1, 20, 1), (1024, 256, 1, 1, 20, 1),
    (512, 512, 3, 1, 20, 6),
]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_conv(B=64, dtype=torch.bfloat16, with_ref=True):
    dev = torch.device("cuda:0")
    rows = []
    tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, ref_fwd=0.0, ref_bwd=0.0, flop=0.0)
    only = os.environ.get("MB_ONLY", "")          # substring of the forward kernel name, e.g. "pp" or "256, 256"
    for (cin, cout, k, s, h, cnt) in V5L:
        p = 2 if k == 6 else k // 2
        kname = ops.kernel_name("fwd", dtype, B, h, h, cin, cout, k, s, p)
        if only and only not in kname and only not in ops.kernel_name("dgrad", dtype, B, h, h, cin, cout, k, s, p):
            continue
        if os.environ.get("MB_K") and int(os.environ["MB_K"]) != k:       # only layers with this kernel size
            continue
        x = torch.randn(B, h, h, cin, device=dev).to(dtype)
        w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(dtype)
        oh, ow = ops.conv_out_hw(h, h, k, s, p)
        dy = torch.randn(B, oh, ow, cout, device=dev).to(dtype)
        wT = ops.weight_transpose(w)
        dw = torch.zeros(cout, k, k, cin, device=dev)
        y = torch.empty(B, oh, ow, cout, device=dev, dtype=dtype)
        dx = torch.empty_like(x)
        flop = 2.0 * B * oh * ow * cout * cin * k * k
        rot = int(os.environ.get("MB_ROTATE", "0"))   # > 0: cycle through that many input / output buffers (defeats the 256 MB MALL)
        if rot > 1:
            xs = [x] + [torch.randn_like(x) for _ in range(rot - 1)]; ys = [y] + [torch.empty_like(y) for _ in range(rot - 1)]
            dys = [dy] + [torch.randn_like(dy) for _ in range(rot - 1)]; dxs = [dx] + [torch.empty_like(dx) for _ in range(rot - 1)]
            ctr = [0]

            def f_rot():
                i = ctr[0] = (ctr[0] + 1) % rot
                ops.conv2d_fwd(xs[i], w, s, p, out=ys[i])

            def d_rot():
                i = ctr[0] = (ctr[0] + 1) % rot
                ops.conv2d_dgrad(dys[i], wT, (h, h), s, p, out=dxs[i])
            t_f = timeit(f_rot, iters=2 * rot)
            t_d = timeit(d_rot, iters=2 * rot) if k != 6 else 0.0
        else:
            t_f = timeit(lambda: ops.conv2d_fwd(x, w, s, p, out=y))
            t_d = timeit(lambda: ops.conv2d_dgrad(dy, wT, (h, h), s, p, out=dx)) if k != 6 else 0.0
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, s, p)) if os.environ.get("MB_WGRAD", "1") == "1" else 1e-9
        r = dict(cin=cin, cout=cout, k=k, s=s, h=h, count=cnt, gflop=flop / 1e9, fwd_kernel=kname,
                 fwd_ms=t_f * 1e3, dgrad_ms=t_d * 1e3, wgrad_ms=t_w * 1e3,
                 fwd_tf=flop / t_f / 1e12, dgrad_tf=(flop / t_d / 1e12 if t_d else 0), wgrad_tf=flop / t_w / 1e12)
        if os.environ.get("MB_FULL", "0") == "1" and s == 1 and k != 6:
            # the Bottleneck.cv1 form of the dgrad: + shortcut-gradient residual + the BatchNorm-backward sums of the producer
            # (reads dy, the residual and the producer's y; writes dx): 4 tensors of traffic
            resid, yprod = torch.randn_like(x), torch.randn_like(x)
            sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
            hand = ops.BnBwdSums(yprod, sc, sh, ops.ACT_SILU)
            t_df = timeit(lambda: ops.conv2d_dgrad(dy, wT, (h, h), s, p, out=dx, residual=resid, bn=hand))
            r.update(dgrad_full_ms=t_df * 1e3, dgrad_full_kernel=ops.kernel_name("dgrad_full", dtype, B, h, h, cin, cout, k, s, p))
            # the teacher's fused forward: folded BN + SiLU (+ residual on the 3x3 of a Bottleneck)
            t_ft = timeit(lambda: ops.conv2d_fwd(x, w, s, p, scale=sc.new_ones(cout), bias=sc.new_zeros(cout), act=ops.ACT_SILU, out=y))
            r.update(fwd_teacher_ms=t_ft * 1e3)
        if with_ref:
            xr = x.permute(0, 3, 1, 2)  # channels_last view
            wr = w.permute(0, 3, 1, 2)
            xr.requires_grad_(True); wr.requires_grad_(True)
            t_rf = timeit(lambda: F.conv2d(xr, wr, stride=s, padding=p))
            yr = F.conv2d(xr, wr, stride=s, padding=p)
            gy = dy.permute(0, 3, 1, 2)
            t_rb = timeit(lambda: torch.autograd.grad(yr, (xr, wr), gy, retain_graph=True))
            r.update(ref_fwd_ms=t_rf * 1e3, ref_bwd_ms=t_rb * 1e3, ref_fwd_tf=flop / t_rf / 1e12,
                     ref_bwd_tf=2 * flop / t_rb / 1e12)
            tot["ref_fwd"] += t_rf * cnt; tot["ref_bwd"] += t_rb * cnt
        tot["fwd"] += t_f * cnt; tot["dgrad"] += t_d * cnt; tot["wgrad"] += t_w * cnt; tot["flop"] += flop * cnt
        rows.append(r)
        print(json.dumps(r), flush=True)
    summ = dict(B=B, dtype=str(dtype), total_fwd_ms=tot["fwd"] * 1e3, total_dgrad_ms=tot["dgrad"] * 1e3,
                total_wgrad_ms=tot["wgrad"] * 1e3, ref_fwd_ms=tot["ref_fwd"] * 1e3, ref_bwd_ms=tot["ref_bwd"] * 1e3,
                fwd_tflops=tot["flop"] / tot["fwd"] / 1e12, gflop_per_img=tot["flop"] / B / 1e9)
    print("SUMMARY " + json.dumps(summ), flush=True)
    return rows, summ


def bench_bn(B=64, dtype=torch.bfloat16):
    """BN+SiLU forward / backward elementwise kernels at every (pixels, channels) of the YOLOv5l step;
    GB/s counts the algorithmic passes: fwd = read y + write z; bwd = 2 reads (reduce) + 2 reads + 1 write (apply)."""
    dev = torch.device("cuda:0")
    tot = dict(fwd=0.0, bwd=0.0, bytes=0.0)
    for (cin, cout, k, s, h, cnt) in V5L:
        ho = h // s
        y = torch.randn(B, ho, ho, cout, device=dev).to(dtype)
        dz = torch.randn(B, ho, ho, cout, device=dev).to(dtype)
        z = torch.empty_like(y); dy = torch.empty_like(y)
        gamma = torch.rand(cout, device=dev) + 0.5
        mean = torch.zeros(cout, device=dev); invstd = torch.ones(cout, device=dev)
        scale = gamma * invstd; shift = -mean * scale
        dg = torch.zeros(cout, device=dev); db = torch.zeros(cout, device=dev)
        t_f = timeit(lambda: ops.bn_act_fwd(y, scale, shift, ops.ACT_SILU, out=z))
        t_b = timeit(lambda: ops.bn_act_bwd(dz, y, gamma, scale, shift, mean, invstd, ops.ACT_SILU, dg, db, out=dy))
        nb = y.numel() * y.element_size()
        r = dict(P=B * ho * ho, C=cout, count=cnt, MB=nb / 1e6, fwd_us=t_f * 1e6, bwd_us=t_b * 1e6,
                 fwd_GBps=2 * nb / t_f / 1e9, bwd_GBps=5 * nb / t_b / 1e9)
        tot["fwd"] += t_f * cnt; tot["bwd"] += t_b * cnt; tot["bytes"] += nb * cnt
        print("BN " + json.dumps(r), flush=True)
    print("BNSUMMARY " + json.dumps(dict(fwd_ms=tot["fwd"] * 1e3, bwd_ms=tot["bwd"] * 1e3, GB_per_pass=tot["bytes"] / 1e9,
                                         fwd_GBps=2 * tot["bytes"] / tot["fwd"] / 1e9,
                                         bwd_GBps=5 * tot["bytes"] / tot["bwd"] / 1e9)), flush=True)


def bench_nms(B=32, A=25200, no=85):
    from efficientteacher_amd.utils.general import nms_ssod_padded
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    pred = torch.zeros(B, A, no)
    pred[..., 0:2] = torch.rand(B, A, 2, generator=g) * 640
    pred[..., 2:4] = torch.rand(B, A, 2, generator=g) * 200 + 4
    pred[..., 4] = torch.rand(B, A, generator=g) ** 16
    pred[..., 5:] = torch.rand(B, A, no - 5, generator=g) ** 4
    pred = pred.to(dev)
    t = timeit(lambda: nms_ssod_padded(pred, 0.1, 0.65), iters=20)
    _, counts, _, ncand = nms_ssod_padded(pred, 0.1, 0.65)
    r = dict(B=B, A=A, no=no, ms=t * 1e3, scan_GBps=pred.numel() * 4 / t / 1e9,
             mean_candidates=float(ncand.float().mean()), mean_kept=float(counts.float().mean()))
    print("NMS " + json.dumps(r), flush=True)
    # val.py configuration of the general NMS: multi_label, conf 0.001, iou 0.65
    from efficientteacher_amd.utils.general import nms_padded
    tv = timeit(lambda: nms_padded(pred, 0.001, 0.65, multi_label=True), iters=5)
    _, cv, _, ncv = nms_padded(pred, 0.001, 0.65, multi_label=True)
    print("NMS_VAL " + json.dumps(dict(B=B, A=A, no=no, ms=tv * 1e3, mean_candidates=float(ncv.float().mean()),
                                       mean_kept=float(cv.float().mean()))), flush=True)
    return r


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "conv"
    print(torch.cuda.get_device_name(0), flush=True)
    if what in ("nms", "all"):
        bench_nms()
    if what in ("bn", "all"):
        bench_bn(B=int(os.environ.get("MB_B", 64)))
    if what in ("conv", "all"):
        bench_conv(B=int(os.environ.get("MB_B", 64)), with_ref=os.environ.get("MB_REF", "1") == "1")
