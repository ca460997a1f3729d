"""Package power and shader clock while ONE kernel of the step runs in a loop (rocm-smi sampled by a thread at ~10 Hz): where the
step's power goes.  python tools/probe/kernel_power.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16


def sampler(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([0-9.]+)", t); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
            if p and s:
                out.append((float(p.group(1)), float(s.group(1))))
        except Exception:
            pass


def run(name, fn, seconds=4.0, flop=0.0, nbytes=0.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    out = out[1:] if len(out) > 2 else out
    P = sum(o[0] for o in out) / max(len(out), 1); S = sum(o[1] for o in out) / max(len(out), 1)
    print(f"{name:46s} {dtm * 1e6:8.1f} us  {flop / dtm / 1e12:7.0f} TFLOP/s {nbytes / dtm / 1e12:5.2f} TB/s   power {P:6.0f} W  sclk {S:5.0f} MHz  ({len(out)} samples)", flush=True)


def conv_case(cin, cout, k, h, B=64):
    p = k // 2
    xs = [torch.randn(B, h, h, cin, device=dev).to(dt) for _ in range(4)]
    w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(dt)
    ys = [torch.empty(B, h, h, cout, device=dev, dtype=dt) for _ in range(4)]
    dw = torch.zeros(cout, k, k, cin, device=dev)
    i = [0]

    def f():
        i[0] = (i[0] + 1) % 4
        ops.conv2d_fwd(xs[i[0]], w, 1, p, out=ys[i[0]])

    def g():
        i[0] = (i[0] + 1) % 4
        ops.conv2d_wgrad(xs[i[0]], ys[i[0]], dw, k, 1, p)
    flop = 2.0 * B * h * h * cin * cout * k * k
    nb = B * h * h * (cin + cout) * 2
    return f, g, flop, nb


time.sleep(2)
stop, out = threading.Event(), []
th = threading.Thread(target=sampler, args=(stop, out)); th.start(); time.sleep(2.5); stop.set(); th.join()
print("idle", out[-3:])
f, g, fl, nb = conv_case(256, 256, 3, 40)
run("conv_gemm_pprs 256->256 3x3 @40 fwd", f, flop=fl, nbytes=nb)
run("conv_wgrad_rs 256->256 3x3 @40", g, flop=fl, nbytes=nb)
f, g, fl, nb = conv_case(1024, 1024, 1, 20)
run("conv_gemm_pp 1024->1024 1x1 @20 fwd", f, flop=fl, nbytes=nb)
f, g, fl, nb = conv_case(128, 128, 3, 80)
run("conv_gemm_rs 128->128 3x3 @80 fwd", f, flop=fl, nbytes=nb)
f, g, fl, nb = conv_case(128, 128, 1, 80)
run("conv1x1_stream 128->128 1x1 @80 fwd", f, flop=fl, nbytes=nb)
run("conv_wgrad_tr 128->128 1x1 @80", g, flop=fl, nbytes=nb)
ys = [torch.randn(64, 40, 40, 256, device=dev).to(dt) for _ in range(4)]
zs = [torch.empty_like(y) for y in ys]
sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
j = [0]


def bn():
    j[0] = (j[0] + 1) % 4
    ops.bn_act_fwd(ys[j[0]], sc, sh, ops.ACT_SILU, out=zs[j[0]])


run("bn_act_fwd 64x40x40x256", bn, nbytes=2 * ys[0].numel() * 2)
a, b = torch.randn(64 * 1024 * 1024, device=dev), torch.empty(64 * 1024 * 1024, device=dev)
run("torch copy 256 MB (HBM read + write)", lambda: b.copy_(a), nbytes=2 * a.numel() * 4)
