"""Seeded random conv problems through whatever kernel the library selects (bf16), element-wise against F.conv2d in fp32 on the
bf16-rounded operands: forward, dgrad and wgrad.  Complements tests/test_conv.py::SELECT (which pins one problem per kernel
instantiation) with shapes nobody chose by hand -- ragged pixel counts, channel counts that are not powers of two, 1-pixel
maps, stride 2 on odd sizes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _problems(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 3]))
        s = int(rng.choice([1, 1, 2]))
        cin = int(rng.choice([8, 16, 24, 64, 72, 128, 192, 256, 320, 512]))
        cout = int(rng.choice([8, 24, 64, 80, 128, 256, 264, 512]))
        h, w = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        n_img = int(rng.integers(1, 5))
        if n_img * h * w * cin * cout * k * k > 6e9:
            continue
        out.append((n_img, h, w, cin, cout, k, s, k // 2))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_problems_gpu(seed):
    from efficientteacher_amd import _lib, ops
    _lib._use_library_for_tests(None, False)
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    seen = set()
    for (N, H, W, Cin, Cout, k, s, p) in _problems(seed, 14):
        x = torch.randn((N, H, W, Cin), generator=g).to(dt).to(dev)
        w = (torch.randn((Cout, k, k, Cin), generator=g) / (k * k * Cin) ** 0.5).to(dt).to(dev)
        OH, OW = ops.conv_out_hw(H, W, k, s, p)
        if OH < 1 or OW < 1:
            continue
        seen.add(ops.kernel_name("fwd", dt, N, H, W, Cin, Cout, k, s, p))
        xr = x.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
        wr = w.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
        ref = F.conv2d(xr, wr, stride=s, padding=p)
        y, _ = ops.conv2d_fwd(x, w, s, p, want_stats=True)
        tol = 2e-2 * max(1.0, ref.abs().max().item())
        assert (y.float().cpu() - ref.detach().permute(0, 2, 3, 1)).abs().max().item() <= tol, (N, H, W, Cin, Cout, k, s)
        dy = torch.randn((N, OH, OW, Cout), generator=g).to(dt).to(dev)
        ref.backward(dy.float().cpu().permute(0, 3, 1, 2))
        dx = ops.conv2d_dgrad(dy, ops.weight_transpose(w), (H, W), s, p)
        dref = xr.grad.permute(0, 2, 3, 1)
        assert (dx.float().cpu() - dref).abs().max().item() <= 2e-2 * max(1.0, dref.abs().max().item()), ("dgrad", N, H, W, Cin, Cout, k, s)
        dw = torch.zeros((Cout, k, k, Cin), dtype=torch.float32, device=dev)
        ops.conv2d_wgrad(x, dy, dw, k, s, p)
        wref = wr.grad.permute(0, 2, 3, 1)
        assert (dw.cpu() - wref).abs().max().item() <= 2e-3 * max(1.0, wref.abs().max().item()), ("wgrad", N, H, W, Cin, Cout, k, s)
    assert len(seen) >= 3, seen


def test_random_problems_emulator(emu):
    """a few of the same generator's small problems on the CPU tier"""
    from efficientteacher_amd import ops
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(7)
    done = 0
    for (N, H, W, Cin, Cout, k, s, p) in _problems(7, 40):
        if N * H * W * Cin * Cout * k * k > 4e7:
            continue
        x = torch.randn((N, H, W, Cin), generator=g).to(dt)
        w = (torch.randn((Cout, k, k, Cin), generator=g) / (k * k * Cin) ** 0.5).to(dt)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=s, padding=p).permute(0, 2, 3, 1)
        y, _ = ops.conv2d_fwd(x, w, s, p, want_stats=True)
        assert (y.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item()), (N, H, W, Cin, Cout, k, s)
        done += 1
        if done == 6:
            break
    assert done >= 4


def test_dgrad_of_a_1x1_stride2_conv(hip):
    """three of the four output-parity classes of a 1x1 stride-2 dgrad are reached by no tap: their gradient is zero (found by
    the random problems above; the model itself has no such layer)"""
    from efficientteacher_amd import ops
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    N, H, W, Cin, Cout = 2, 7, 9, 16, 24
    w = (torch.randn((Cout, 1, 1, Cin), generator=g) / Cin ** 0.5).to(dt)
    OH, OW = ops.conv_out_hw(H, W, 1, 2, 0)
    dy = torch.randn((N, OH, OW, Cout), generator=g).to(dt)
    xr = torch.zeros((N, Cin, H, W), requires_grad=True)
    F.conv2d(xr, w.float().permute(0, 3, 1, 2), stride=2).backward(dy.float().permute(0, 3, 1, 2))
    dx = ops.conv2d_dgrad(hip.t(dy), ops.weight_transpose(hip.t(w)), (H, W), 2, 0)
    ref = xr.grad.permute(0, 2, 3, 1)
    assert (dx.float().cpu() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert (dx.float().cpu()[:, 1::2] == 0).all() and (dx.float().cpu()[:, :, 1::2] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [10, 11, 12, 13])
def test_random_epilogues_gpu(seed):
    """random problems with random epilogue options: folded scale / bias, SiLU / ReLU, residual, output written into a channel
    slice of a wider buffer; dgrad with accumulate or the shortcut residual; the packed-image stem at odd sizes"""
    from efficientteacher_amd import _lib, ops
    _lib._use_library_for_tests(None, False)
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    probs = _problems(seed, 8) + [(int(rng.integers(1, 4)), int(rng.integers(6, 70)), int(rng.integers(6, 200)), 8, int(rng.choice([16, 32, 48, 64])), 6, 2, 2)]
    for (N, H, W, Cin, Cout, k, s, p) in probs:
        OH, OW = ops.conv_out_hw(H, W, k, s, p)
        if OH < 1 or OW < 1:
            continue
        x = torch.randn((N, H, W, Cin), generator=g).to(dt).to(dev)
        w = (torch.randn((Cout, k, k, Cin), generator=g) / (k * k * Cin) ** 0.5).to(dt).to(dev)
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), stride=s, padding=p).permute(0, 2, 3, 1)
        act = int(rng.choice([ops.ACT_NONE, ops.ACT_SILU, ops.ACT_RELU]))
        use_sc, use_res, use_slice = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        sc = (torch.rand(Cout, generator=g) + 0.5).to(dev) if use_sc else None
        bi = torch.randn(Cout, generator=g).to(dev) if use_sc else None
        res = torch.randn((N, OH, OW, Cout), generator=g).to(dt).to(dev) if use_res else None
        exp = ref * sc.cpu() + bi.cpu() if use_sc else ref.clone()
        exp = F.silu(exp) if act == ops.ACT_SILU else F.relu(exp) if act == ops.ACT_RELU else exp
        if use_res:
            exp = exp + res.float().cpu()
        if use_slice:
            wide = torch.zeros((N, OH, OW, Cout + 24), dtype=dt, device=dev)
            ops.conv2d_fwd(x, w, s, p, scale=sc, bias=bi, act=act, residual=res, out=wide[..., 16:16 + Cout])
            y = wide[..., 16:16 + Cout]
            assert (wide[..., :16] == 0).all() and (wide[..., 16 + Cout:] == 0).all()
        else:
            y = ops.conv2d_fwd(x, w, s, p, scale=sc, bias=bi, act=act, residual=res)
        tag = (N, H, W, Cin, Cout, k, s, act, use_sc, use_res, use_slice)
        assert (y.float().cpu() - exp).abs().max().item() <= 3e-2 * max(1.0, exp.abs().max().item()), tag
        if k == 6 or s != 1:
            continue
        dy = torch.randn((N, OH, OW, Cout), generator=g).to(dt).to(dev)
        xr = torch.zeros((N, Cin, H, W), requires_grad=True)
        F.conv2d(xr, w.float().cpu().permute(0, 3, 1, 2), stride=s, padding=p).backward(dy.float().cpu().permute(0, 3, 1, 2))
        dref = xr.grad.permute(0, 2, 3, 1)
        r2 = torch.randn((N, H, W, Cin), generator=g).to(dt).to(dev)
        dx = ops.conv2d_dgrad(dy, ops.weight_transpose(w), (H, W), s, p, residual=r2)
        assert (dx.float().cpu() - (dref + r2.float().cpu())).abs().max().item() <= 3e-2 * max(1.0, dref.abs().max().item()) + 3e-2 * r2.float().abs().max().item(), ("dgrad+res",) + tag
