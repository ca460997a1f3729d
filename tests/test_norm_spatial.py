"""BN+SiLU fwd/bwd, SPPF max-pool, nearest upsample, input packing vs plain PyTorch fp32 references."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _mk(hip, shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(hip.device)


def _tol(dtype):
    return 3e-5 if dtype == torch.float32 else 3e-2


@pytest.fixture(params=["single-block finalize", "distributed finalize"])
def finalize_form(request):
    """the statistics finalize has two forms (csrc/norm.hip): one block per 16 channels for few partial rows (the default below
    2048 rows), fp64 atomics + ticket over many blocks above; ET_BN_FIN_SMALL=0 forces the second on these small tensors"""
    import os
    old = os.environ.get("ET_BN_FIN_SMALL")
    if request.param.startswith("distributed"):
        os.environ["ET_BN_FIN_SMALL"] = "0"
    yield request.param
    if old is None:
        os.environ.pop("ET_BN_FIN_SMALL", None)
    else:
        os.environ["ET_BN_FIN_SMALL"] = old


# the last shape gives every thread 6 or 7 vectors (512 blocks for 819200 / 1638400 vectors): the paired loop AND its odd tail
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 7, 9, 32), (1, 5, 5, 48), (3, 4, 4, 8), (2, 40, 40, 24), (4, 72, 72, 8), (2, 160, 160, 128)])
def test_bn_silu_fwd_bwd(hip, finalize_form, shape, dtype):
    from efficientteacher_amd import ops
    N, H, W, C = shape
    # conv in front so that the stats come from the conv epilogue exactly as in the model
    x = _mk(hip, (N, H, W, 16), dtype, 1)
    w = _mk(hip, (C, 1, 1, 16), dtype, 2, 0.3)
    y, stats = ops.conv2d_fwd(x, w, 1, 0, want_stats=True)
    gamma = _mk(hip, (C,), torch.float32, 3, 0.2) + 1
    beta = _mk(hip, (C,), torch.float32, 4, 0.1)
    rm = torch.zeros(C, device=hip.device)
    rv = torch.ones(C, device=hip.device)
    res = _mk(hip, shape, dtype, 5)
    scale, shift, mean, invstd = ops.bn_finalize(stats, N * H * W, gamma, beta, 1e-3, 0.03, rm, rv)
    z = ops.bn_act_fwd(y, scale, shift, ops.ACT_SILU, residual=res)
    # reference: fp32 on the conv output as the kernel saw it (fp32 accumulators)
    yr = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2)).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.copy_(gamma.cpu()); bn.bias.copy_(beta.cpu())
    bn.train()
    zr = F.silu(bn(yr)) + res.float().cpu().permute(0, 3, 1, 2)
    err = (z.float().cpu().permute(0, 3, 1, 2) - zr).abs().max().item()
    assert err <= _tol(dtype) * 4, err
    assert torch.allclose(rm.cpu(), bn.running_mean, rtol=1e-3, atol=2e-3 if dtype != torch.float32 else 1e-6)
    assert torch.allclose(rv.cpu(), bn.running_var, rtol=5e-3 if dtype != torch.float32 else 1e-4, atol=1e-5)
    # backward
    dz = _mk(hip, shape, dtype, 6)
    dg = torch.zeros(C, device=hip.device); db = torch.zeros(C, device=hip.device)
    dy = ops.bn_act_bwd(dz, y, gamma, scale, shift, mean, invstd, ops.ACT_SILU, dg, db)
    zr.backward(dz.float().cpu().permute(0, 3, 1, 2))
    ref = yr.grad
    err = (dy.float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err <= _tol(dtype) * max(1.0, ref.abs().max().item()) * 2, err
    assert torch.allclose(dg.cpu(), bn.weight.grad, rtol=_tol(dtype) * 10, atol=_tol(dtype) * 20)
    assert torch.allclose(db.cpu(), bn.bias.grad, rtol=_tol(dtype) * 10, atol=_tol(dtype) * 20)
    # eval affine
    sc, sh = ops.bn_eval_affine(gamma, beta, rm, rv, 1e-3)
    bn.eval()
    ze = ops.bn_act_fwd(y, sc, sh, ops.ACT_SILU)
    zre = F.silu(bn(yr.detach()))
    assert (ze.float().cpu().permute(0, 3, 1, 2) - zre).abs().max().item() <= _tol(dtype) * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 9, 7, 16), (1, 20, 20, 32), (1, 3, 2, 8), (3, 20, 20, 72), (1, 24, 24, 40), (1, 34, 33, 8)])
def test_sppf_pool_chain(hip, shape, dtype):
    """x -> y1 -> y2 -> y3 written into slices of one concat buffer; backward through the chain (the pooled maps have plateaus:
    the gradient routing checks the first-maximum tie rule against torch).  Shapes include channel counts that are not a multiple of 64
    (72, 40) and maps beyond SPPF's 20 x 20."""
    from efficientteacher_amd import ops
    N, H, W, C = shape
    cat = torch.zeros((N, H, W, 4 * C), dtype=dtype, device=hip.device)
    x = _mk(hip, (N, H, W, C), dtype, 11)
    cat[..., :C].copy_(x)
    idx = []
    for i in range(3):
        _, ix = ops.maxpool5_fwd(cat[..., i * C:(i + 1) * C], out=cat[..., (i + 1) * C:(i + 2) * C])
        idx.append(ix)
    xr = x.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    y1 = F.max_pool2d(xr, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    ref = torch.cat([xr, y1, y2, y3], 1)
    assert torch.equal(cat.float().cpu().permute(0, 3, 1, 2), ref.detach())
    dcat = _mk(hip, (N, H, W, 4 * C), dtype, 12)
    d2 = ops.maxpool5_bwd(dcat[..., 3 * C:], idx[2], base=dcat[..., 2 * C:3 * C])
    d1 = ops.maxpool5_bwd(d2, idx[1], base=dcat[..., C:2 * C])
    dx = ops.maxpool5_bwd(d1, idx[0], base=dcat[..., :C])
    ref.backward(dcat.float().cpu().permute(0, 3, 1, 2))
    err = (dx.float().cpu().permute(0, 3, 1, 2) - xr.grad).abs().max().item()
    assert err <= _tol(dtype) * 8, err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_upsample_and_pack(hip, dtype):
    from efficientteacher_amd import ops
    x = _mk(hip, (2, 3, 5, 16), dtype, 21)
    buf = torch.zeros((2, 6, 10, 40), dtype=dtype, device=hip.device)
    ops.upsample2x_fwd(x, out=buf[..., 8:24])
    ref = F.interpolate(x.float().cpu().permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    assert torch.equal(buf[..., 8:24].float().cpu().permute(0, 3, 1, 2), ref)
    assert (buf[..., :8] == 0).all() and (buf[..., 24:] == 0).all()
    dy = _mk(hip, (2, 6, 10, 16), dtype, 22)
    dx = ops.upsample2x_bwd(dy)
    xr = x.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    F.interpolate(xr, scale_factor=2, mode="nearest").backward(dy.float().cpu().permute(0, 3, 1, 2))
    assert (dx.float().cpu().permute(0, 3, 1, 2) - xr.grad).abs().max().item() <= _tol(dtype) * 4
    img = torch.rand(2, 3, 6, 4).to(hip.device)
    p = ops.pack_input(img, dtype)
    assert p.shape == (2, 6, 4, 8) and (p[..., 3:] == 0).all()
    assert torch.equal(p[..., :3].float().cpu(), img.cpu().permute(0, 2, 3, 1).to(dtype).float())
