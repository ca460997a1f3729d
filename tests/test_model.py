"""Tiny-width SSOD detector (width 0.125, depth 0.33, 64x64 input) vs the reference's own outputs
(tests/golden/model_tiny.npz, produced by oracle/make_golden.py from models/detector/yolo_ssod.py):
state_dict key compatibility, eval forward (decode), train forward, ComputeLoss, full backward.
fp32 parity mode: 1e-4 on losses / outputs (BASELINE north_star tolerance), gradients 2e-3 relative.
"""
import numpy as np
import pytest
import torch

from tests.conftest import golden

YAML = "efficientteacher_amd/configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml"


def build(hip, dtype=torch.float32):
    import os
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from tests.conftest import ROOT
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33])
    cfg.freeze()
    g = golden("model_tiny")
    model = Model(cfg)
    sd = {k[3:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("w__")}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model = model.to(hip.device)
    model.set_compute_dtype(dtype)
    return cfg, model, g


def test_state_dict_keys_and_init_match_reference(hip):
    """Same module names / construction order => same keys AND the same default init for seed 0."""
    import os
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from tests.conftest import ROOT
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33])
    g = golden("model_tiny")
    torch.manual_seed(0)
    m = Model(cfg)
    keys = {k[3:].replace("__", ".") for k in g.files if k.startswith("w__")}
    assert set(m.state_dict().keys()) == keys
    # conv weights are untouched by make_golden's BN perturbation: default init must be identical
    w = m.state_dict()["backbone.stage2_2.m.0.cv2.conv.weight"].numpy()
    assert np.array_equal(w, g["w__backbone__stage2_2__m__0__cv2__conv__weight"])
    assert np.allclose(m.head.anchors.numpy(), g["anchors"]) and np.allclose(m.stride.numpy(), g["stride"])
    assert np.array_equal(m.state_dict()["head.m.0.bias"].numpy(), g["w__head__m__0__bias"])


def test_eval_forward(hip):
    cfg, model, g = build(hip)
    model.eval()
    with torch.no_grad():
        (z, xs), feats = model(hip.t(g["x"]))
    assert z.shape == g["eval_z"].shape
    err = np.abs(z.cpu().numpy() - g["eval_z"]).max()
    assert err <= 1e-4 * max(1.0, np.abs(g["eval_z"]).max()), err
    for i in range(3):
        assert np.abs(xs[i].cpu().numpy() - g[f"eval_x{i}"]).max() <= 1e-4
        assert feats[i].shape == g[f"eval_feat{i}"].shape
        assert np.abs(feats[i].cpu().numpy() - g[f"eval_feat{i}"]).max() <= 1e-4


def test_train_forward_loss_backward(hip):
    from efficientteacher_amd.models.loss import ComputeLoss
    cfg, model, g = build(hip)
    model.train()
    closs = ComputeLoss(model, cfg)
    pred, feats = model(hip.t(g["x"]))
    for i in range(3):
        assert tuple(pred[i].shape) == g[f"train_p{i}"].shape
        assert np.abs(pred[i].detach().cpu().numpy() - g[f"train_p{i}"]).max() <= 2e-4
    loss, items = closs(pred, hip.t(g["targets"]))
    assert abs(loss.item() - float(g["train_loss"][0])) <= 1e-4 * max(1.0, abs(float(g["train_loss"][0])))
    got = np.array([items[k].item() for k in ("box", "obj", "cls")])
    assert np.allclose(got, g["train_items"], rtol=1e-4, atol=1e-5)
    model.zero_grad()
    loss.backward()
    # running statistics updated exactly like torch's BatchNorm (momentum 0.03, unbiased variance)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("b__"):
            name = k[3:].replace("__", ".")
            assert np.allclose(sd[name].cpu().numpy(), g[k], rtol=1e-4, atol=1e-5), name
    worst = 0.0
    for name, p in model.named_parameters():
        ref = g["g__" + name.replace(".", "__")]
        if name.startswith("det_"):
            continue        # netD: zero-weighted DA loss -> no gradient (documented in yolo_ssod.py)
        got = p.grad.detach().cpu().numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(got - ref).max() / scale
        worst = max(worst, err)
        assert err <= 2e-3, (name, err)
    print("worst relative grad error", worst)


@pytest.mark.parametrize("width,dtype", [(0.125, torch.float32), (0.125, torch.bfloat16), (0.5, torch.bfloat16), (0.5, torch.float16)])
def test_eval_c3_in_place_shortcut_equals_unfused_path(hip, width, dtype, monkeypatch):
    """in the teacher's fused C3 stem the LAST Bottleneck writes cv2's output over buf[..., :c_] while that very slice is
    its shortcut operand (et_conv2d_fwd with residual == y, same pixel stride: every lane loads the element it is about to store).
    The contract is stated in include/et_hip.h; here the fused eval forward is compared with the path that keeps the shortcut in a
    separate tensor, on widths that put Bottleneck.cv2 on every tile kind that carries a residual (per-tap 128-row tiles at width
    0.125; row-shift 128x64 / 128x128 tiles and the 256x256 ping-pong tile at width 0.5, depth 0.33 = one Bottleneck per C3)."""
    import os
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.backbone import common
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from tests.conftest import ROOT
    if width > 0.125 and hip.emulated:
        pytest.skip("the wide model runs on the GPU tier (minutes in the emulator)")
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(["Model.width_multiple", width, "Model.depth_multiple", 0.33])
    cfg.freeze()
    torch.manual_seed(3)
    model = Model(cfg).to(hip.device)
    model.set_compute_dtype(dtype)
    with torch.no_grad():                       # running statistics away from (0, 1) so that the folded affine is not trivial
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    model.eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(4)).to(hip.device)
    with torch.no_grad():
        (z_fused, _), _ = model(x)
        monkeypatch.setattr(common, "c3_stem_fusable", lambda *a, **k: False)
        (z_plain, _), _ = model(x)
    scale = z_plain.abs().max().item()
    err = (z_fused - z_plain).abs().max().item()
    assert err <= (1e-5 if dtype == torch.float32 else 2e-2) * scale, (err, scale)


def test_bf16_mode_close_to_fp32(hip):
    """Performance mode (bf16 storage, fp32 accumulate) stays close to the fp32 golden outputs."""
    cfg, model, g = build(hip, torch.bfloat16)
    model.eval()
    with torch.no_grad():
        (z, xs), _ = model(hip.t(g["x"]))
    rel = np.abs(z.cpu().numpy() - g["eval_z"]).max() / np.abs(g["eval_z"]).max()
    assert rel <= 3e-2, rel


@pytest.mark.parametrize("group,stale", [(2, 1000), (3, 1), (16, 2)])
def test_grouped_wgrad_queue_matches_ungrouped(hip, group, stale):
    """ops.WgradQueue: any grouping / staleness policy must give the same parameter gradients as launching every
    weight gradient on its own (the bf16 path is the one that groups; compare in bf16 against group = 1)."""
    from efficientteacher_amd import ops
    from efficientteacher_amd.models.loss import ComputeLoss
    grads = []
    q = ops.WGRAD_QUEUE
    saved = (q.group, q.stale)
    try:
        for g_, s_ in ((1, 1000), (group, stale)):
            q.group, q.stale = g_, s_
            cfg, model, g = build(hip, torch.bfloat16)
            model.train()
            closs = ComputeLoss(model, cfg)
            pred, _ = model(hip.t(g["x"]))
            loss, _ = closs(pred, hip.t(g["targets"]))
            model.zero_grad()
            loss.backward()
            assert not q.pending                              # everything was flushed by the end of backward
            grads.append({k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()})
    finally:
        q.group, q.stale = saved
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.allclose(a, b, rtol=2e-2, atol=2e-3 * max(1.0, a.abs().max().item())), k


def test_split_batch_fallback_paths(hip):
    """autograd.split_batch: halves whose gradient does NOT come from the fused loss (or is missing) are stitched by
    the copy fallback."""
    from efficientteacher_amd.autograd import split_batch
    p = torch.randn(4, 3, 5, 5, 8, device=hip.device, requires_grad=True)
    a, b = split_batch(p, 1)
    (a * 2.0).sum().backward()                                # only the first half is used
    ref = torch.zeros_like(p); ref[:1] = 2.0
    assert torch.equal(p.grad, ref)
    p.grad = None
    a, b = split_batch(p, 3)
    ((a * 1.5).sum() + (b * -1.0).sum()).backward()           # both halves, ordinary torch gradients
    ref = torch.full_like(p, 1.5); ref[3:] = -1.0
    assert torch.equal(p.grad, ref)


@pytest.mark.parametrize("B,H,W", [(2, 96, 160), (1, 160, 32)])
def test_rectangular_inputs_match_the_oracle(hip, B, H, W):
    """val.py feeds rectangular letter-boxed batches (multiples of the 32-pixel stride): eval outputs and a train-mode
    forward / backward against the oracle model with the same weights, fp32 mode.  Exercises ragged tile grids in the stem,
    odd pyramid sizes (5 x 1 at stride 32) and the upsample / concat paths off the square case."""
    from oracle import model as o_model
    cfg, model, g = build(hip, torch.float32)
    ref = o_model.Model.from_cfg(cfg)
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    rng = np.random.default_rng(B * 1000 + H)
    x = rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32)
    model.eval(); ref.eval()
    with torch.no_grad():
        (z, _), _ = model(hip.t(x))
        zr = ref(torch.from_numpy(x))[0][0]
    assert z.shape == zr.shape
    assert (z.cpu() - zr).abs().max().item() <= 1e-4 * max(1.0, zr.abs().max().item())
    model.train(); ref.train()
    out, _ = model(hip.t(x))
    outr = ref(torch.from_numpy(x))[0]
    loss = sum((o.float() ** 2).mean() for o in out)
    lossr = sum((o ** 2).mean() for o in outr)
    assert abs(loss.item() - lossr.item()) <= 1e-4 * abs(lossr.item())
    model.zero_grad(); loss.backward(); lossr.backward()
    for k in ("backbone.stage1.conv.weight", "neck.C2.cv3.conv.weight", "head.m.2.bias"):
        a = dict(model.named_parameters())[k].grad.cpu()
        b = dict(ref.named_parameters())[k].grad
        assert (a - b).abs().max().item() <= 2e-3 * max(b.abs().max().item(), 1e-6), k


def test_two_consumer_gradients_are_merged_in_kernels(hip):
    """The six tensors of the YOLOv5 graph with two consumers (backbone C3 / C4, the neck's two lateral outputs and its P3 / P4
    outputs): the consumer whose backward runs last adds its gradient into the other's inside its own kernel (autograd.GradFork:
    dgrad epilogue `accumulate`, et_upsample2x_bwd `accumulate`) -- no torch `add` of gradient branches is left in the step
    (VERDICT r02 item 8).  Counts the merges of one backward and checks that none fell back to the sum."""
    from efficientteacher_amd import autograd as ag
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from efficientteacher_amd.models.loss import ComputeLoss
    import os
    from tests.conftest import ROOT
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33])
    cfg.freeze()
    torch.manual_seed(0)
    model = Model(cfg).to(hip.device).train()
    counts = {"merged": 0, "summed": 0}
    orig = ag._ForkFn.backward

    def spy(ctx, ga, gb):
        if ga is not None and gb is not None:
            counts["merged" if ctx.holder.merged else "summed"] += 1
        return orig(ctx, ga, gb)
    ag._ForkFn.backward = staticmethod(spy)
    try:
        x = hip.t(np.random.default_rng(0).random((2, 3, 64, 64), dtype=np.float32))
        pred, _ = model(x)
        loss, _ = ComputeLoss(model, cfg)(pred, hip.t(np.array([[0, 3, .5, .5, .3, .4], [1, 7, .4, .6, .2, .2]], np.float32)))
        loss.backward()
    finally:
        ag._ForkFn.backward = orig
    assert counts == {"merged": 6, "summed": 0}, counts
