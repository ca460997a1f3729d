"""FlatSGD + the three EMA classes vs the reference's torch.optim.SGD / ModelEMA / CosineEMA
(tests/golden/optim.npz: two optimizer steps with stored gradients on the tiny model)."""
import numpy as np
import torch

from tests.conftest import golden
from tests.test_model import build

KEYS = ["backbone.stage1.conv.weight", "backbone.stage1.bn.weight", "head.m.0.bias", "neck.C1.cv3.conv.weight"]


def test_sgd_and_ema_two_steps(hip):
    from efficientteacher_amd.optim import FlatSGD
    from efficientteacher_amd.utils.torch_utils import CosineEMA, ModelEMA
    cfg, model, _ = build(hip, torch.bfloat16)
    g = golden("optim")
    model.train()
    params = dict(model.named_parameters())
    for k in KEYS:
        assert np.array_equal(params[k].detach().cpu().numpy(), g["p0__" + k.replace(".", "__")])
    opt = FlatSGD(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
    assert [len(x["params"]) for x in opt.param_groups] == [int(g["groups"][2]), int(g["groups"][1]), int(g["groups"][0])]
    ema = ModelEMA(model)
    semi = CosineEMA(ema.ema, decay_start=0.999, decay_end=0.9999, total_epoch=300)
    for step in range(2):
        opt.zero_grad()
        for k in KEYS:
            params[k].grad.copy_(hip.t(g[f"g{step}__" + k.replace(".", "__")]))
        opt.step()
        ema.update(model)
        semi.update(ema.ema)
    esd, ssd = ema.ema.state_dict(), semi.ema.state_dict()
    for k in KEYS:
        kk = k.replace(".", "__")
        assert np.allclose(params[k].detach().cpu().numpy(), g["p2__" + kk], rtol=1e-6, atol=1e-8), k
        assert np.allclose(esd[k].cpu().numpy(), g["ema2__" + kk], rtol=1e-6, atol=1e-8), k
        assert np.allclose(ssd[k].cpu().numpy(), g["semi2__" + kk], rtol=1e-6, atol=1e-8), k
    # the bf16 shadow the MFMA kernels read follows the fp32 master
    f = model.flat_state()
    o, n = f.w_range
    assert torch.equal(f.shadow.float().cpu(), f.params[o:o + n].to(torch.bfloat16).float().cpu())
    assert ema.updates == 2 and abs(ema.decay(2) - 0.9999 * (1 - np.exp(-2 / 2000))) < 1e-12


def test_ema_teacher_is_independent_copy(hip):
    from efficientteacher_amd.utils.torch_utils import ModelEMA
    cfg, model, g = build(hip)
    ema = ModelEMA(model)
    assert not ema.ema.training and all(not p.requires_grad for p in ema.ema.parameters())
    a, b = model.flat_state().params, ema.ema.flat_state().params
    assert a.data_ptr() != b.data_ptr() and torch.equal(a.cpu(), b.cpu())
    model.flat_state().params.add_(1.0)
    assert not torch.equal(a.cpu(), b.cpu())
    with torch.no_grad():
        (z, _), _ = ema.ema(hip.t(g["x"]))
    assert np.abs(z.cpu().numpy() - g["eval_z"]).max() <= 1e-4 * np.abs(g["eval_z"]).max()


def test_adamw_matches_torch(hip):
    """FlatAdamW (cfg.adam, reference trainer.py:210-217) against torch.optim.AdamW with the reference's three groups on the
    same parameters and gradients: three steps, incl. the default 0.01 decay the reference leaves on biases / BN weights"""
    from efficientteacher_amd.optim import FlatAdamW
    cfg, model, _ = build(hip, torch.bfloat16)
    model.train()
    opt = FlatAdamW(model, lr=0.01, betas=(0.937, 0.999), weight_decay=5e-4)
    import torch.nn as nn
    g_bnw, g_w, g_b = [], [], []
    clones = {}
    for v in model.modules():
        for name, grp in (("bias", g_b), ("weight", g_bnw if isinstance(v, nn.BatchNorm2d) else g_w)):
            p = getattr(v, name, None)
            if isinstance(p, nn.Parameter):
                c = p.detach().cpu().clone().requires_grad_(True)
                clones[id(p)] = c
                grp.append(c)
    ref = torch.optim.AdamW(g_b, lr=0.01, betas=(0.937, 0.999))
    ref.add_param_group({'params': g_w, 'weight_decay': 5e-4})
    ref.add_param_group({'params': g_bnw})
    gen = torch.Generator().manual_seed(5)
    params = list(model.parameters())
    for step in range(3):
        opt.zero_grad()
        for p in params:
            gr = torch.randn(p.shape, generator=gen) * 0.1
            p.grad.copy_(gr.to(hip.device))
            clones[id(p)].grad = gr.clone()
        opt.step()
        ref.step()
    worst = max(((p.detach().cpu() - clones[id(p)].detach()).abs().max() / (clones[id(p)].detach().abs().max() + 1e-6)).item()
                for p in params)
    assert worst <= 2e-6, worst
    sd = opt.state_dict()
    opt2 = FlatAdamW(model, lr=0.01, betas=(0.937, 0.999), weight_decay=5e-4)
    opt2.load_state_dict(sd)
    assert opt2.steps == 3 and torch.equal(opt2.exp_avg, opt.exp_avg)


def test_cast_f32_to_lp_bf16_is_torch_rounding(hip):
    """et_cast_f32_to_lp with a bf16 destination (the bf16 weight shadow): round-to-nearest-even like torch, for aligned arenas (eight elements per
    thread + a scalar tail) and for an unaligned slice (scalar kernel)"""
    from efficientteacher_amd import ops
    g = torch.Generator().manual_seed(3)
    src = (torch.randn(4099, generator=g) * torch.logspace(-20, 20, 4099)).to(hip.device)
    src[5] = float("inf"); src[6] = -0.0; src[7] = 1.00390625          # a tie: rounds to even
    for sl in (slice(0, 4099), slice(0, 4096), slice(1, 1000), slice(8, 13)):
        s = src[sl]
        d = torch.empty(s.numel(), dtype=torch.bfloat16, device=hip.device)
        ops.cast_f32_to_lp(s, d)
        assert torch.equal(d.cpu(), s.cpu().to(torch.bfloat16)), sl


def test_scale_cast_bf16_vector_and_tail(hip):
    """et_scale_cast to bf16 (the loss gradients handed to the head's backward): src * scale [* dev_scale], rounded like torch,
    for a length that is not a multiple of eight (vector kernel + scalar tail) and an unaligned slice"""
    from efficientteacher_amd import ops
    g = torch.Generator().manual_seed(4)
    src = torch.randn(2051, generator=g).to(hip.device)
    dev_scale = torch.tensor([0.37], device=hip.device)
    for s in (src, src[3:1500]):
        out = ops.scale_cast(s, torch.bfloat16, scale=1.7, dev_scale=dev_scale)
        ref = (s.cpu() * (torch.tensor(1.7) * dev_scale.cpu()[0])).to(torch.bfloat16)
        assert torch.equal(out.cpu(), ref)
        out2 = ops.scale_cast(s, torch.bfloat16, scale=-2.0)
        assert torch.equal(out2.cpu(), (s.cpu() * -2.0).to(torch.bfloat16))
