"""Isolated launches of the step's MFMA-bound conv shapes under controlled neighbours (where does the
isolated -> in-step gap of conv_gemm_pprs_kernel / conv_gemm_pp_kernel come from?).

    ISO_MODE=hot|rot|nbr  python tools/probe/iso_conv.py          (one mode per process: rocprofv3 --pmc groups by kernel + grid)

  hot : the same operand buffers every launch, launches back to back (what tools/microbench.py measures by default: inputs and outputs
        stay in the 256 MB memory-side cache)
  rot : 6 operand sets in rotation (> 256 MB in total: every launch reads HBM-cold activations; the weights stay the same)
  nbr : rot + the in-step neighbours: the launch's INPUT is produced just before it by the BatchNorm normalise pass of the previous
        layer (bn_act_fwd: HBM-bound, leaves z MALL-warm, the chip in its HBM-bound power state) and a different layer's weights are
        used by every launch (cold weights in L2: 6 weight sets)
Prints one JSON line per (shape, direction): mean HIP-event duration per launch in that mode.  Under rocprofv3 the same process is
summarised by tools/pmc_by_grid.py (counters and durations per kernel + grid)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops  # noqa: E402

# (Cin, Cout, k, s, H, B): the dominant 3x3 bucket, the 20x20 3x3 bucket, the deep 1x1 (plain GEMM), the 128-channel 3x3
SHAPES = [(256, 256, 3, 1, 40, 64), (512, 512, 3, 1, 20, 64), (1024, 1024, 1, 1, 20, 64), (512, 512, 1, 1, 40, 64), (128, 128, 3, 1, 80, 64),
          (256, 256, 3, 1, 40, 32), (512, 512, 3, 1, 20, 32)]


def main():
    mode = os.environ.get("ISO_MODE", "hot")
    iters = int(os.environ.get("ISO_ITERS", "24"))
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    nset = 1 if mode == "hot" else 6
    for (cin, cout, k, s, h, B) in SHAPES:
        p = k // 2
        xs = [torch.randn(B, h, h, cin, device=dev).to(dt) for _ in range(nset)]
        raw = [torch.randn(B, h, h, cin, device=dev).to(dt) for _ in range(nset)] if mode == "nbr" else None
        nw = nset if mode == "nbr" else 1
        ws = [(torch.randn(cout, k, k, cin, device=dev) * 0.05).to(dt) for _ in range(nw)]
        wTs = [ops.weight_transpose(w) for w in ws]
        ys = [torch.empty(B, h, h, cout, device=dev, dtype=dt) for _ in range(nset)]
        dys = [torch.randn(B, h, h, cout, device=dev).to(dt) for _ in range(nset)]
        rawd = [torch.randn(B, h, h, cout, device=dev).to(dt) for _ in range(nset)] if mode == "nbr" else None
        dxs = [torch.empty(B, h, h, cin, device=dev, dtype=dt) for _ in range(nset)]
        sc_i, sh_i = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
        sc_o, sh_o = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        for direction in ("fwd", "dgrad"):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for it in range(-3, iters):
                i = it % nset
                if mode == "nbr":       # the producer of the operand, as in the step (forward: the previous layer's normalise pass)
                    if direction == "fwd":
                        ops.bn_act_fwd(raw[i], sc_i, sh_i, ops.ACT_SILU, out=xs[i])
                    else:
                        ops.bn_act_fwd(rawd[i], sc_o, sh_o, ops.ACT_SILU, out=dys[i])
                if it >= 0:
                    ev[it][0].record()
                if direction == "fwd":
                    ops.conv2d_fwd(xs[i], ws[i % nw], s, p, out=ys[i], want_stats=False)
                else:
                    ops.conv2d_dgrad(dys[i], wTs[i % nw], (h, h), s, p, out=dxs[i])
                if it >= 0:
                    ev[it][1].record()
            torch.cuda.synchronize()
            us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            flop = 2.0 * B * h * h * cout * cin * k * k
            mean = sum(us) / len(us)
            print(json.dumps(dict(mode=mode, shape=[cin, cout, k, s, h, B], dir=direction,
                                  kernel=ops.kernel_name(direction, dt, B, h, h, cin, cout, k, s, p),
                                  us_mean=round(mean, 2), us_median=round(us[len(us) // 2], 2), us_min=round(us[0], 2),
                                  tflops=round(flop / mean / 1e6, 1))), flush=True)
        del xs, ys, dys, dxs, ws, wTs, raw, rawd
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
