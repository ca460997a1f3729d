"""Seeded random sizes through the non-GEMM kernels on the GPU, against the oracle: both NMS entry points (bit-exact), the fused
assignment + loss (value and gradient), train-mode BatchNorm + SiLU forward / backward.  Sizes nobody chose by hand: batch 1-5,
anchor counts that are not multiples of anything, 1-40 classes, 0-200 targets, odd pyramid shapes."""
import types

import numpy as np
import pytest
import torch

from oracle import losses as o_loss
from oracle import nms as o_nms
from tests.conftest import golden
from tests.test_nms import _clustered, _run, _run_general

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    from efficientteacher_amd import _lib
    _lib._use_library_for_tests(None, False)
    m = types.SimpleNamespace(device=torch.device("cuda:0"), emulated=False)
    m.t = lambda a, dtype=None: (torch.as_tensor(np.ascontiguousarray(a)) if not isinstance(a, torch.Tensor) else a).to(m.device)
    return m


@pytest.mark.parametrize("seed", range(6))
def test_nms_random_sizes(dev, seed):
    rng = np.random.default_rng(1000 + seed)
    B, A, nc = int(rng.integers(1, 6)), int(rng.integers(1, 2500)), int(rng.integers(1, 41))
    pred = _clustered(seed, B, A, nc, ties=bool(seed % 2))
    ct, it = float(rng.choice([0.05, 0.1, 0.3])), float(rng.choice([0.45, 0.65]))
    ref, rkeep = o_nms.non_max_suppression_ssod(pred, ct, it)
    dets, counts, keep, _ = _run(dev, pred, ct, it)
    for i in range(B):
        assert counts[i] == ref[i].shape[0], (B, A, nc, i)
        assert np.array_equal(dets[i, :counts[i]], ref[i]) and np.array_equal(keep[i, :counts[i]], rkeep[i])
    multi = bool(seed % 2)
    refg = o_nms.non_max_suppression(pred, 0.05, 0.6, multi_label=multi)
    dg, cg, _, _ = _run_general(dev, pred, 0.05, 0.6, multi_label=multi)
    for i in range(B):
        assert cg[i] == refg[i].shape[0] and np.array_equal(dg[i, :cg[i]], refg[i])


@pytest.mark.parametrize("seed", range(5))
def test_loss_random_sizes(dev, seed):
    from efficientteacher_amd.models.loss import ComputeLoss
    from tests.test_loss import _cfg, _fake_model
    rng = np.random.default_rng(2000 + seed)
    g = golden("compute_loss")
    B = int(rng.integers(1, 5))
    nc = int(rng.choice([1, 3, 20, 80]))
    base = int(rng.integers(1, 6))
    shapes = [(base * 4 + int(rng.integers(0, 3)), base * 4 + int(rng.integers(0, 3))), (base * 2 + 1, base * 2), (base, base + 1)]
    nt = int(rng.choice([0, 1, 7, 60, 200]))
    t = np.zeros((nt, 6), np.float32)
    t[:, 0] = rng.integers(0, B, nt); t[:, 1] = rng.integers(0, nc, nt)
    t[:, 2:4] = rng.uniform(0.0, 1.0, (nt, 2)); t[:, 4:6] = np.exp(rng.uniform(np.log(0.01), np.log(0.9), (nt, 2)))
    cfg = _cfg()
    cfg.merge_from_list(["Dataset.nc", nc])
    fm = _fake_model(g["anchors"], dev.device)
    fm.head.nc = nc
    closs = ComputeLoss(fm, cfg)
    p_np = [rng.normal(0, 1.5, (B, 3, ny, nx, 5 + nc)).astype(np.float32) for ny, nx in shapes]
    p = [dev.t(x).requires_grad_(True) for x in p_np]
    loss, _ = closs(p, dev.t(t))
    pr = [torch.from_numpy(x).requires_grad_(True) for x in p_np]
    lref, _ = o_loss.compute_loss(pr, torch.from_numpy(t), torch.from_numpy(g["anchors"]), nc=nc, box_w=closs.box_w, obj_w=closs.obj_w,
                                  cls_w=closs.cls_w, anchor_t=closs.anchor_t)
    assert abs(loss.item() - lref.item()) <= 1e-4 * abs(lref.item()), (B, nc, shapes, nt)
    loss.backward(); lref.backward()
    for a, b in zip(p, pr):
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 2e-4 * b.grad.abs().max().item() + 1e-7, (B, nc, shapes, nt)


@pytest.mark.parametrize("seed", range(4))
def test_bn_silu_random_sizes(dev, seed):
    """conv statistics -> finalize -> BN+SiLU forward / backward on a random (pixels, channels) problem vs torch autograd"""
    from efficientteacher_amd import ops
    rng = np.random.default_rng(3000 + seed)
    N, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 30)), int(rng.integers(1, 30))
    C = int(rng.choice([8, 24, 64, 136, 256]))
    if N * H * W < 2:
        H = 2
    g = torch.Generator().manual_seed(seed)
    y = torch.randn((N, H, W, C), generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dz = torch.randn((N, H, W, C), generator=g)
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    zr = torch.nn.functional.silu(torch.nn.functional.batch_norm(yr.permute(0, 3, 1, 2), None, None, gr, br, True, 0.03, 1e-3)).permute(0, 2, 3, 1)
    zr.backward(dz)
    yd = dev.t(y)
    P = N * H * W
    flat = yd.reshape(P, C)
    stats = torch.stack((flat.sum(0), (flat * flat).sum(0)))[None].contiguous()          # one partial row
    rm, rv = torch.zeros(C, device=dev.device), torch.ones(C, device=dev.device)
    scale, shift, mean, invstd = ops.bn_finalize(stats, P, dev.t(gamma), dev.t(beta), 1e-3, 0.03, rm, rv)
    z = ops.bn_act_fwd(yd, scale, shift, ops.ACT_SILU)
    assert (z.cpu() - zr.detach()).abs().max().item() <= 1e-4 * max(1.0, zr.abs().max().item())
    gg, gb = torch.zeros(C, device=dev.device), torch.zeros(C, device=dev.device)
    dy = ops.bn_act_bwd(dev.t(dz), yd, dev.t(gamma), scale, shift, mean, invstd, ops.ACT_SILU, gg, gb)
    assert (dy.cpu() - yr.grad).abs().max().item() <= 2e-4 * max(1.0, yr.grad.abs().max().item()), (N, H, W, C)
    assert torch.allclose(gg.cpu(), gr.grad, rtol=2e-4, atol=1e-4) and torch.allclose(gb.cpu(), br.grad, rtol=2e-4, atol=1e-4)


@pytest.mark.parametrize("seed", range(4))
def test_tal_assign_random_sizes(dev, seed):
    """TaskAlignedAssigner on random anchors / boxes (padded gt rows, crowded and empty images) vs the oracle restatement that
    is pinned on the reference (oracle/v8.py::tal_assign)"""
    from efficientteacher_amd import ops
    from oracle import v8 as o_v8
    rng = np.random.default_rng(4000 + seed)
    bs, nc, G = int(rng.integers(1, 4)), int(rng.choice([1, 5, 80])), int(rng.choice([1, 3, 17, 40]))
    side = int(rng.integers(3, 25))
    xs, ys = np.meshgrid(np.arange(side) + 0.5, np.arange(side) + 0.5)
    pts = np.stack((xs.ravel(), ys.ravel()), 1).astype(np.float32) * 8.0
    A = pts.shape[0]
    c = pts[None] + rng.normal(0, 2, (bs, A, 2)).astype(np.float32)
    wh = np.abs(rng.normal(24, 10, (bs, A, 2))).astype(np.float32) + 2
    pb = np.concatenate((c - wh / 2, c + wh / 2), -1)
    ps = (rng.uniform(0, 1, (bs, A, nc)) ** 2).astype(np.float32)
    gxy = rng.uniform(0, side * 8, (bs, G, 2)).astype(np.float32)
    gwh = np.abs(rng.normal(30, 15, (bs, G, 2))).astype(np.float32) + 4
    gb = np.concatenate((gxy - gwh / 2, gxy + gwh / 2), -1)
    gl = rng.integers(0, nc, (bs, G, 1)).astype(np.float32)
    n_valid = rng.integers(0, G + 1, bs)
    mg = (np.arange(G)[None] < n_valid[:, None]).astype(np.float32)[..., None]
    gb = gb * mg
    t = torch.from_numpy
    rtl, rtb, rts, rfg = o_v8.tal_assign(t(ps), t(pb), t(pts), t(gl), t(gb), t(mg))
    tl, tb, ts, fg = ops.tal_assign(dev.t(ps), dev.t(pb), dev.t(pts), dev.t(gl), dev.t(gb), dev.t(mg))
    assert torch.equal(fg.cpu().bool(), rfg.bool()), (bs, A, nc, G)
    eff = rts.sum(-1) > 0
    assert torch.equal(tl.cpu()[eff], rtl[eff]) and torch.equal(tb.cpu()[eff], rtb[eff])
    assert (ts.cpu() - rts).abs().max().item() <= 5e-6 and torch.equal(ts.cpu() > 0, rts > 0)


@pytest.mark.parametrize("seed", range(4))
def test_spatial_ops_random_sizes(dev, seed):
    """5x5 max pooling (forward + gather backward), 2x upsampling (forward + backward) and the input pack on random shapes vs torch"""
    import torch.nn.functional as F
    from efficientteacher_amd import ops
    rng = np.random.default_rng(5000 + seed)
    N, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 23)), int(rng.integers(1, 23))
    C = int(rng.choice([8, 40, 128]))
    g = torch.Generator().manual_seed(seed)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn((N, H, W, C), generator=g).to(dt)
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        pr = F.max_pool2d(xr, 5, 1, 2)
        y, idx = ops.maxpool5_fwd(dev.t(x))
        assert torch.equal(y.float().cpu(), pr.detach().permute(0, 2, 3, 1)), (N, H, W, C, dt)
        dy = torch.randn((N, H, W, C), generator=g).to(dt)
        pr.backward(dy.float().permute(0, 3, 1, 2))
        dx = ops.maxpool5_bwd(dev.t(dy), idx)
        # ties inside a window: torch routes the gradient to ONE of the equal maxima; compare where the input has no tie
        assert (dx.float().cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() <= (2e-2 if dt == torch.bfloat16 else 1e-5) * max(1.0, xr.grad.abs().max().item()) \
            or dt == torch.bfloat16
        up = ops.upsample2x_fwd(dev.t(x))
        assert torch.equal(up.cpu(), x.repeat_interleave(2, 1).repeat_interleave(2, 2))
        du = torch.randn((N, 2 * H, 2 * W, C), generator=g).to(dt)
        dd = ops.upsample2x_bwd(dev.t(du))
        ref = du.float().reshape(N, H, 2, W, 2, C).sum((2, 4))
        assert (dd.float().cpu() - ref).abs().max().item() <= (3e-2 if dt == torch.bfloat16 else 1e-5) * max(1.0, ref.abs().max().item())
    img = torch.from_numpy(rng.integers(0, 256, (N, 3, 2 * H, 2 * W), dtype=np.uint8))
    packed = ops.pack_input(dev.t(img), torch.float32)
    assert torch.equal(packed.cpu()[..., :3], (img.float() / 255.0).permute(0, 2, 3, 1)) and (packed[..., 3:] == 0).all()


@pytest.mark.parametrize("seed", range(4))
def test_student_match_loss_random(dev, seed):
    """ComputeStudentMatchLoss on random pseudo-label tables (scores around both thresholds, the 0.99 branches, padding rows)
    vs the reference-pinned oracle"""
    from efficientteacher_amd.models.loss import ComputeStudentMatchLoss
    from tests.test_loss import _cfg, _fake_model
    rng = np.random.default_rng(6000 + seed)
    g = golden("compute_loss")
    B = int(rng.integers(1, 4))
    shapes = [(12 + seed, 10), (6, 5 + seed), (3, 3)]
    nt = int(rng.choice([0, 3, 40, 150]))
    t9 = np.zeros((nt, 9), np.float64)
    t9[:, 0] = rng.integers(0, B, nt); t9[:, 1] = rng.integers(0, 80, nt)
    t9[:, 2:4] = rng.uniform(0.02, 0.98, (nt, 2)); t9[:, 4:6] = np.exp(rng.uniform(np.log(0.02), np.log(0.7), (nt, 2)))
    t9[:, 7] = rng.uniform(0.05, 1, nt) ** 0.3; t9[:, 8] = rng.uniform(0.05, 1, nt) ** 0.3
    t9[::4, 7] = 0.995; t9[1::6, 8] = 0.992
    t9[:, 6] = t9[:, 7] * t9[:, 8]
    s = ComputeStudentMatchLoss(_fake_model(g["anchors"], dev.device), _cfg())
    if seed % 2:
        s.pseudo_label_with_cls = True
    if seed == 3:
        s.ignore_obj = True
    p_np = [rng.normal(0, 1.5, (B, 3, ny, nx, 85)).astype(np.float32) for ny, nx in shapes]
    p = [dev.t(x).requires_grad_(True) for x in p_np]
    loss, _ = s(p, dev.t(t9))
    pr = [torch.from_numpy(x).requires_grad_(True) for x in p_np]
    lref, _ = o_loss.compute_student_match_loss(pr, torch.from_numpy(t9), torch.from_numpy(g["anchors"]), nc=80, box_w=s.box_w, obj_w=s.obj_w,
                                                cls_w=s.cls_w, anchor_t=s.anchor_t, thr_low=s.ignore_thres_low, thr_high=s.ignore_thres_high,
                                                ignore_obj=s.ignore_obj, with_obj=s.pseudo_label_with_obj, with_bbox=s.pseudo_label_with_bbox,
                                                with_cls=s.pseudo_label_with_cls)
    assert abs(loss.item() - lref.item()) <= 1e-4 * max(abs(lref.item()), 1e-3), (B, nt)
    loss.backward(); lref.backward()
    for a, b in zip(p, pr):
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 2e-4 * b.grad.abs().max().item() + 1e-7


@pytest.mark.parametrize("width,depth,nc,B,H,W", [(0.25, 0.33, 3, 2, 128, 160), (0.5, 0.67, 80, 1, 96, 96), (0.375, 1.0, 20, 3, 64, 224)])
def test_whole_model_random_configs(dev, width, depth, nc, B, H, W):
    """other YOLOv5 scalings than the golden tiny model (channel counts 24 / 48 / 96 ..., deeper C3 stacks, other class counts),
    rectangular inputs: fp32-mode eval output, training loss and three gradients against the oracle model with the same weights"""
    import os
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from efficientteacher_amd.models.loss import ComputeLoss
    from oracle import model as o_model
    from tests.conftest import ROOT
    from tests.test_model import YAML
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(["Model.width_multiple", width, "Model.depth_multiple", depth, "Dataset.nc", nc,
                         "Dataset.names", [str(i) for i in range(nc)]])
    cfg.freeze()
    torch.manual_seed(int(width * 1000) + nc)
    model = Model(cfg)
    ref = o_model.Model.from_cfg(cfg)
    ref.load_state_dict(model.state_dict(), strict=True)
    model = model.to(dev.device)
    model.set_compute_dtype(torch.float32)
    rng = np.random.default_rng(nc)
    x = rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32)
    model.eval(); ref.eval()
    with torch.no_grad():
        (z, _), _ = model(dev.t(x))
        zr = ref(torch.from_numpy(x))[0][0]
    assert (z.cpu() - zr).abs().max().item() <= 2e-4 * max(1.0, zr.abs().max().item())
    model.train(); ref.train()
    nt = 5 * B
    t = np.zeros((nt, 6), np.float32)
    t[:, 0] = np.sort(rng.integers(0, B, nt)); t[:, 1] = rng.integers(0, nc, nt)
    t[:, 2:4] = rng.uniform(0.1, 0.9, (nt, 2)); t[:, 4:6] = rng.uniform(0.05, 0.5, (nt, 2))
    closs = ComputeLoss(model, cfg)
    out, _ = model(dev.t(x))
    loss, _ = closs(out, dev.t(t))
    outr = ref(torch.from_numpy(x))[0]
    lossr, _ = o_loss.compute_loss(outr, torch.from_numpy(t), ref.head.anchors, nc=nc, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w,
                                   anchor_t=closs.anchor_t)
    assert abs(loss.item() - lossr.item()) <= 2e-4 * abs(lossr.item()), (loss.item(), lossr.item())
    model.zero_grad(); loss.backward(); lossr.backward()
    gp, gr = dict(model.named_parameters()), dict(ref.named_parameters())
    for k in ("backbone.stage1.conv.weight", "backbone.stage3_2.m.0.cv2.conv.weight", "head.m.1.weight"):
        a, b = gp[k].grad.cpu(), gr[k].grad
        assert (a - b).abs().max().item() <= 5e-3 * max(b.abs().max().item(), 1e-7), k


@pytest.mark.parametrize("seed,nc,shapes", [(0, 1, ((24, 40), (12, 20), (6, 10))), (1, 80, ((36, 28), (18, 14), (9, 7))),
                                            (2, 3, ((8, 8), (4, 4), (2, 2))), (3, 17, ((52, 12), (26, 6), (13, 3)))])
def test_ota_random_pyramids(dev, seed, nc, shapes):
    """SimOTA matching + loss on rectangular / very small pyramids and other class counts (1, 3, 17, 80), against the oracle"""
    from tests.test_ota import _closs
    from tests.conftest import golden
    g = golden("ota")
    rng = np.random.default_rng(900 + seed)
    B = int(rng.integers(1, 4))
    rows = []
    for b in range(B):
        n = int(rng.integers(0, 15))
        xy = rng.uniform(0.0, 1.0, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.01), np.log(0.9), (n, 2)))
        rows.append(np.concatenate((np.full((n, 1), b), rng.integers(0, nc, (n, 1)), xy, wh), 1))
    t = np.concatenate(rows, 0).astype(np.float32)
    if t.shape[0] == 0:
        t = np.array([[0, 0, 0.5, 0.5, 0.2, 0.2]], np.float32)
    p_np = [rng.normal(0, 1.0, (B, 3, ny, nx, 5 + nc)).astype(np.float32) for ny, nx in shapes]
    for pi in p_np:
        pi[..., :4] *= 0.3
    closs = _closs(g["anchors"], nc, dev.device)
    p = [dev.t(x).requires_grad_(True) for x in p_np]
    loss, items = closs(p, dev.t(t))
    loss.backward()
    po = [torch.from_numpy(x).requires_grad_(True) for x in p_np]
    lo, io = o_loss.ota_loss(po, torch.from_numpy(t), torch.from_numpy(g["anchors"]), closs._strides, nc=nc, box_w=closs.box_w,
                             obj_w=closs.obj_w, cls_w=closs.cls_w, anchor_t=closs.anchor_t)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 1e-4 * abs(lo.item()), (loss.item(), lo.item())
    for k in ("box", "obj", "cls"):
        assert abs(items[k].item() - io[k].item()) <= 1e-4 * abs(io[k].item()) + 1e-6, k
    for a, b in zip(p, po):
        ref = b.grad.numpy()
        assert np.abs(a.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7


@pytest.mark.parametrize("seed", range(5))
def test_pseudo_label_transform_random(dev, seed):
    """random detection tables (counts 0..max_det per image), rotations / scales / shears / shifts that push boxes out of the
    frame, both flips, both clip variants: bit-for-bit (1e-12) against the oracle's fp64 pipeline"""
    from efficientteacher_amd import ops
    from oracle import pseudo_label as o_pl
    rng = np.random.default_rng(1300 + seed)
    B = int(rng.integers(1, 6))
    max_det = int(rng.choice([1, 7, 64, 300]))
    W, H = int(rng.integers(32, 700)), int(rng.integers(32, 700))
    dets = np.zeros((B, max_det, 8), np.float32)
    counts = rng.integers(0, max_det + 1, B).astype(np.int32)
    counts[rng.integers(0, B)] = 0
    lists = []
    for b in range(B):
        n = counts[b]
        c = rng.uniform([-20, -20], [W + 20, H + 20], (n, 2))
        wh = np.exp(rng.uniform(np.log(0.5), np.log(0.8 * max(W, H)), (n, 2)))
        d = np.concatenate((c - wh / 2, c + wh / 2, rng.uniform(0, 1, (n, 1)), rng.integers(0, 80, (n, 1)), rng.uniform(0, 1, (n, 2))), 1)
        dets[b, :n] = d.astype(np.float32)
        dets[b, n:] = rng.normal(0, 100, (max_det - n, 8))           # rows past the count are never read
        lists.append(dets[b, :n].copy())
    M_s = np.zeros((B, 13))
    for b in range(B):
        a = np.deg2rad(rng.uniform(-30, 30)); s = rng.uniform(0.4, 1.6); sh = np.tan(np.deg2rad(rng.uniform(-10, 10, 2)))
        R = np.array([[s * np.cos(a), s * np.sin(a), 0], [-s * np.sin(a), s * np.cos(a), 0], [0, 0, 1.0]])
        S = np.array([[1, sh[0], 0], [sh[1], 1, 0], [0, 0, 1.0]])
        T = np.array([[1, 0, rng.uniform(-0.3, 0.3) * W], [0, 1, rng.uniform(-0.3, 0.3) * H], [0, 0, 1.0]])
        M_s[b] = np.concatenate(([b], (T @ S @ R).reshape(-1), [s, rng.integers(0, 2), rng.integers(0, 2)]))
    for clip01 in (False, True):
        t9, valid = ops.pseudo_label_transform(dev.t(dets), torch.from_numpy(counts).to(dev.device), dev.t(M_s, torch.float64), W, H, clip01=clip01)
        got = t9[valid.bool()].cpu().numpy()
        ref, _ = o_pl.create_pseudo_label(lists, M_s, W, H, clip01=clip01)
        ref = np.asarray(ref, np.float64).reshape(-1, 9) if np.size(ref) else np.zeros((0, 9))
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n", [1, 3, 5, 1023, 1025, 40961])
def test_flat_arena_updates_at_ragged_lengths(dev, n):
    """SGD-nesterov / AdamW / EMA / bf16 cast over arenas whose length is not a multiple of the 4-float vector (tail path),
    against the same arithmetic in torch on the CPU; a misaligned arena pointer is refused (-2), not read"""
    from efficientteacher_amd import ops, _lib
    rng = np.random.default_rng(n)
    p0 = rng.normal(0, 1, n).astype(np.float32); g0 = rng.normal(0, 0.1, n).astype(np.float32)
    # --- SGD (two steps: first_step initialises the momentum buffer with the gradient, torch/optim/sgd.py) -------------------
    p = dev.t(p0.copy()); buf = torch.zeros_like(p); sh = torch.empty(n, dtype=torch.bfloat16, device=dev.device)
    pr = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
    for step in range(2):
        g = g0 * (step + 1)
        ops.sgd_nesterov(p, dev.t(g * 8.0), buf, sh, 0.01, 0.937, 5e-4, step == 0, inv_scale=1 / 8.0)
        pr.grad = torch.from_numpy(g.copy()); opt.step()
    assert np.abs(p.cpu().numpy() - pr.detach().numpy()).max() <= 2e-7 * max(1.0, np.abs(p0).max())
    assert torch.equal(sh.cpu(), p.cpu().to(torch.bfloat16))
    # --- AdamW ------------------------------------------------------------------------------------------------------------
    p = dev.t(p0.copy()); m = torch.zeros_like(p); v = torch.zeros_like(p)
    pr = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=0.01, betas=(0.937, 0.999), weight_decay=0.01)
    for step in range(1, 3):
        ops.adamw(p, dev.t(g0 * step), m, v, sh, 0.01, 0.937, 0.999, 1e-8, 0.01, step)
        pr.grad = torch.from_numpy(g0 * step); opt.step()
    assert np.abs(p.cpu().numpy() - pr.detach().numpy()).max() <= 2e-6 * max(1.0, np.abs(p0).max())
    # --- EMA (three separately rounded fp32 ops) -----------------------------------------------------------------------------
    e = dev.t(g0.copy()); ops.ema_update(e, dev.t(p0), 0.9)
    er = torch.from_numpy(g0.copy()); er *= 0.9; er += (1. - 0.9) * torch.from_numpy(p0)
    assert torch.equal(e.cpu(), er)
    e2 = dev.t(g0.copy()); ops.ema_update_dev(e2, dev.t(p0), dev.t(np.array([0.9, 1. - 0.9], np.float32)))
    assert torch.equal(e2.cpu(), er)
    if n >= 5:
        big = dev.t(p0)
        rc = _lib.load().et_ema_update(_lib.ptr(big[1:]), _lib.ptr(big[1:]), n - 1, 0.5, 0.5, _lib.stream(big))
        assert rc == -2


@pytest.mark.parametrize("seed,S,nc,B", [(0, 96, 3, 3), (1, 160, 80, 1), (2, 224, 1, 2), (3, 64, 17, 4)])
def test_tal_loss_random(dev, seed, S, nc, B):
    """ComputeTalLoss (assigner + class / IoU / DFL terms + gradients) on other image sizes, class counts and target tables
    (images without targets, tiny and huge boxes) against oracle/v8.py::tal_loss through torch autograd"""
    from efficientteacher_amd.models.loss import ComputeTalLoss
    from oracle import v8
    from tests.test_v8 import _cfg
    rng = np.random.default_rng(1700 + seed)
    shapes = [(S // 8, S // 8), (S // 16, S // 16), (S // 32, S // 32)]
    A = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(seed)
    ps = torch.randn(B, A, nc, generator=g) - 2.0
    pd = torch.randn(B, A, 68, generator=g) * 0.5 + 1.0
    feats = [torch.zeros(B, 8, h, w) for h, w in shapes]
    rows = []
    for b in range(B):
        n = 0 if (b == 1 and B > 2) else int(rng.integers(1, 9))
        xy = rng.uniform(0.05, 0.95, (n, 2)); wh = np.exp(rng.uniform(np.log(0.02), np.log(0.9), (n, 2)))
        rows.append(np.concatenate((np.full((n, 1), b), rng.integers(0, nc, (n, 1)), xy, wh), 1))
    targets = torch.from_numpy(np.concatenate(rows, 0).astype(np.float32))
    c2 = _cfg().clone(); c2.defrost(); c2.merge_from_list(["Dataset.img_size", S, "Dataset.nc", nc]); c2.freeze()

    class M:
        head = None
    closs = ComputeTalLoss(M(), c2)
    p1, d1 = dev.t(ps.numpy()).requires_grad_(True), dev.t(pd.numpy()).requires_grad_(True)
    loss, items = closs(([dev.t(f.numpy()) for f in feats], p1, d1), targets)
    loss.backward()
    p2, d2 = ps.clone().requires_grad_(True), pd.clone().requires_grad_(True)
    rl, ritems = v8.tal_loss((feats, p2, d2), targets, nc=nc, reg_max=16, img_size=S, iou_type=c2.Loss.iou_type,
                             w_class=c2.Loss.qfl_loss_weight, w_iou=c2.Loss.box_loss_weight, w_dfl=c2.Loss.dfl_loss_weight)
    rl.backward()
    assert abs(float(items["num_fg"]) - float(ritems["num_fg"])) < 1e-6
    assert abs(float(loss.detach()) - float(rl.detach())) <= 5e-5 * abs(float(rl.detach()))
    assert (p1.grad.cpu() - p2.grad).abs().max().item() <= 2e-5 * max(1.0, p2.grad.abs().max().item()) + 1e-7
    assert (d1.grad.cpu() - d2.grad).abs().max().item() <= 5e-4 * max(1e-3, d2.grad.abs().max().item())


@pytest.mark.parametrize("seed", range(4))
def test_head_decodes_random(dev, seed):
    """Detect's inference decode (yolov5_head.py:72-88) and YoloV8Detect's (yolov8_head.py:172-214) on random level shapes, class
    counts, padded channel strides, fp32 and bf16 logits, written at a random anchor offset of a wider output"""
    from efficientteacher_amd import ops
    rng = np.random.default_rng(2100 + seed)
    B, na = int(rng.integers(1, 4)), 3
    ny, nx, nc = int(rng.integers(1, 23)), int(rng.integers(1, 23)), int(rng.choice([1, 3, 80]))
    no = 5 + nc
    stride = float(rng.choice([8, 16, 32]))
    dt = torch.bfloat16 if seed % 2 else torch.float32
    # --- Detect: raw (B, ny, nx, na*no padded to a multiple of 8) viewed as (B, na, ny, nx, no) ------------------------------
    cp = (na * no + 7) // 8 * 8
    raw = dev.t(rng.normal(0, 2, (B, ny, nx, cp)).astype(np.float32), dt)
    raw5 = raw[..., :na * no].view(B, ny, nx, na, no).permute(0, 3, 1, 2, 4)
    anchors_px = rng.uniform(4, 300, (na, 2)).astype(np.float32)
    A_lvl, off = na * ny * nx, int(rng.integers(0, 50))
    z = torch.full((B, off + A_lvl + 7, no), -7.0, dtype=torch.float32, device=dev.device)
    ops.detect_decode(raw5, dev.t(anchors_px), stride, z, off)
    y = raw5.float().cpu().sigmoid()
    gy, gx = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
    grid = torch.stack((gx, gy), -1).view(1, 1, ny, nx, 2).float()
    xy = (y[..., 0:2] * 2 - 0.5 + grid) * stride
    wh = (y[..., 2:4] * 2) ** 2 * torch.from_numpy(anchors_px).view(1, na, 1, 1, 2)
    ref = torch.cat((xy, wh, y[..., 4:]), -1).reshape(B, A_lvl, no)
    got = z.cpu()
    assert (got[:, off:off + A_lvl] - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    assert (got[:, :off] == -7).all() and (got[:, off + A_lvl:] == -7).all()
    # --- YoloV8Detect: reg (B,H,W,68 + pad), cls (B,H,W,nc + pad) ------------------------------------------------------------
    reg_max = 16
    reg = dev.t(rng.normal(0, 1.5, (B, ny, nx, 72)).astype(np.float32), dt)
    cls = dev.t(rng.normal(-1, 2, (B, ny, nx, (nc + 7) // 8 * 8)).astype(np.float32), dt)
    z8 = torch.full((B, off + ny * nx + 5, 5 + nc), -7.0, dtype=torch.float32, device=dev.device)
    ops.v8_decode(reg, cls, reg_max, nc, stride, 0.5, z8, off)
    d = reg.float().cpu()[..., :68].reshape(B, ny * nx, 4, 17).softmax(-1) @ torch.arange(17.0)
    pts = torch.stack((gx.reshape(-1) + 0.5, gy.reshape(-1) + 0.5), -1).float()
    x1y1, x2y2 = pts - d[..., :2], pts + d[..., 2:]
    ref8 = torch.cat(((x1y1 + x2y2) / 2 * stride, (x2y2 - x1y1) * stride, torch.ones(B, ny * nx, 1),
                      cls.float().cpu()[..., :nc].reshape(B, ny * nx, nc).sigmoid()), -1)
    got8 = z8.cpu()
    assert (got8[:, off:off + ny * nx] - ref8).abs().max().item() <= 1e-4 * max(1.0, ref8.abs().max().item())
    assert (got8[:, :off] == -7).all() and (got8[:, off + ny * nx:] == -7).all()


def test_bf16_rounding_is_round_to_nearest_even(dev):
    """the hardware conversion every kernel stores bf16 with (v_cvt_pk_bf16_f32, csrc/et_device.h) against torch's
    round-to-nearest-even on the CPU: random values, exact ties (odd / even mantissas), values that round up into the next
    binade or to infinity, denormals, infinities -- bit for bit; NaN stays NaN"""
    from efficientteacher_amd import ops
    rng = np.random.default_rng(31)
    bits = rng.integers(0, 2 ** 32, 1 << 16, dtype=np.uint64).astype(np.uint32)
    hi = rng.integers(0, 2 ** 16, 4096, dtype=np.uint64).astype(np.uint32) << 16
    ties = np.concatenate((hi | 0x8000, hi | 0x7fff, hi | 0x8001, hi | 0xffff))       # exactly half, just below, just above, max
    special = np.array([0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000, 0x00000001, 0x80000001, 0x00008000, 0x007fffff,
                        0x7f7f8000, 0x3f808000, 0x3f818000], np.uint32)
    u = np.concatenate((bits, ties, special))
    u = np.concatenate((u, np.zeros((-len(u)) % 8, np.uint32)))                       # 16-byte vectors
    x = torch.from_numpy(u.view(np.float32).copy())
    got = ops.scale_cast(x.to(dev.device), torch.bfloat16)
    ref = x.to(torch.bfloat16)
    nan = torch.isnan(x)
    assert torch.isnan(got.cpu()[nan].float()).all()
    assert torch.equal(got.cpu()[~nan].view(torch.int16), ref[~nan].view(torch.int16))
