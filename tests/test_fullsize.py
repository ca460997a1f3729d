"""Size-independent properties checked at BASELINE.json's FULL sizes (YOLOv5l, 640x640, 32+32 images per GPU),
where the oracle would take too long: adjointness of the three conv kernels, linearity, BatchNorm invariants,
NMS invariants (sortedness, no surviving overlap, idempotence), EMA / SGD closed forms, pseudo-label round trip.
GPU only (the emulator runs the same kernels at test sizes in the other files)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    from efficientteacher_amd import _lib
    if not torch.cuda.is_available():
        pytest.fail("-m gpu selected but no GPU is visible")
    _lib._use_library_for_tests(None, False)
    _lib.load()
    return torch.device("cuda:0")


def _rnd(shape, dev, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


@pytest.mark.parametrize("cin,cout,k,s,h", [(256, 256, 3, 1, 40), (128, 256, 3, 2, 80), (512, 512, 1, 1, 20),
                                            (64, 64, 3, 1, 160), (256, 128, 1, 1, 80), (128, 128, 3, 1, 80)])
def test_conv_adjoint_and_linearity_full_size(dev, cin, cout, k, s, h):
    """<conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(x, dy)>  (bf16 operands, fp32 accumulation) and
    conv(2*x1 - x2) == 2*conv(x1) - conv(x2) up to bf16 rounding, at B = 64."""
    from efficientteacher_amd import ops
    B, p = 64, k // 2
    x = _rnd((B, h, h, cin), dev, 1)
    w = _rnd((cout, k, k, cin), dev, 2, 0.05)
    oh, ow = ops.conv_out_hw(h, h, k, s, p)
    dy = _rnd((B, oh, ow, cout), dev, 3)
    y = ops.conv2d_fwd(x, w, s, p)
    dx = ops.conv2d_dgrad(dy, ops.weight_transpose(w), (h, h), s, p)
    dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=dev)
    ops.conv2d_wgrad(x, dy, dw, k, s, p)
    a = (y.double() * dy.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    nrm = (y.double().norm() * dy.double().norm()).item()
    assert abs(a - c) <= 2e-3 * nrm, (a, c, nrm)        # y and dx are rounded to bf16 on store, dw is fp32
    assert abs(b - c) <= 2e-3 * nrm, (b, c, nrm)
    x2 = _rnd((B, h, h, cin), dev, 4)
    lhs = ops.conv2d_fwd((2 * x.float() - x2.float()).to(torch.bfloat16), w, s, p).float()
    rhs = 2 * y.float() - ops.conv2d_fwd(x2, w, s, p).float()
    err = (lhs - rhs).abs().max().item()
    assert err <= 0.05 * max(1.0, rhs.abs().max().item()), err


@pytest.mark.parametrize("cin,cout,k,s,h", [(64, 64, 3, 1, 160), (128, 128, 3, 1, 80), (256, 256, 3, 1, 40), (1024, 1024, 1, 1, 20), (128, 128, 1, 1, 80)])
def test_forward_and_dgrad_repeat_bit_equal_full_size(dev, cin, cout, k, s, h):
    """idempotence at B = 64: forward and dgrad launches have no atomics, so 25 launches on the same operands give 25 bit-equal outputs.
    (r06: the property that exposes a staging race.  With the buffer-descriptor pieces conv_gemm_rs_kernel<128, 64> failed it on the 160 x 160
    maps in most processes -- the single-barrier LDS rings handed a slot back to the DMA while a slower wave's reads of it were still
    queued; since the waits in front of those barriers include lgkmcnt(0): 0 of 540 launches differ, profiles/r06_lds_ring_war_race.txt)"""
    from efficientteacher_amd import ops
    B, p = 64, k // 2
    x = _rnd((B, h, h, cin), dev, 1)
    w = _rnd((cout, k, k, cin), dev, 2, 0.05)
    ref = ops.conv2d_fwd(x, w, s, p)
    dy = _rnd(tuple(ref.shape), dev, 3)
    wT = ops.weight_transpose(w)
    dref = ops.conv2d_dgrad(dy, wT, (h, h), s, p)
    for _ in range(25):
        assert torch.equal(ops.conv2d_fwd(x, w, s, p), ref)
        assert torch.equal(ops.conv2d_dgrad(dy, wT, (h, h), s, p), dref)


def test_bn_invariants_full_size(dev):
    """train-mode BN over (64, 80, 80, 128): the normalised output has per-channel mean 0 / variance 1, and the
    backward satisfies sum(dy) = 0 and sum(dy * xhat) = 0 per channel (identity activation)."""
    from efficientteacher_amd import ops
    B, H, C = 64, 80, 128
    x = _rnd((B, H, H, 64), dev, 5)
    w = _rnd((C, 1, 1, 64), dev, 6, 0.2)
    y, stats = ops.conv2d_fwd(x, w, 1, 0, want_stats=True)
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev)
    scale, shift, mean, invstd = ops.bn_finalize(stats, B * H * H, gamma, beta, 1e-3, 0.03)
    yf = y.float().reshape(-1, C)
    assert torch.allclose(mean, yf.mean(0), atol=2e-3)
    assert torch.allclose(invstd, 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-3), rtol=2e-3)
    z = ops.bn_act_fwd(y, scale, shift, ops.ACT_NONE).float().reshape(-1, C)
    xhat = (z - beta) / gamma
    assert xhat.mean(0).abs().max().item() < 2e-2 and (xhat.var(0, unbiased=False) - 1).abs().max().item() < 3e-2
    dz = _rnd((B, H, H, C), dev, 7)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dyb = ops.bn_act_bwd(dz, y, gamma, scale, shift, mean, invstd, ops.ACT_NONE, dg, db).float().reshape(-1, C)
    n = dyb.shape[0]
    xh = (yf - mean) * invstd
    assert (dyb.sum(0).abs() / n).max().item() < 2e-3
    assert ((dyb * xh).sum(0).abs() / n).max().item() < 2e-3
    assert torch.allclose(db, dz.float().reshape(-1, C).sum(0), rtol=1e-3, atol=1.0)


def _iou(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ar = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    return inter / (ar(a)[:, None] + ar(b)[None, :] - inter)


@pytest.mark.parametrize("val_path", [False, True])
def test_nms_invariants_full_size(dev, val_path):
    """(32, 25200, 85): scores sorted, every row above the threshold, no same-class pair above the IoU threshold
    among the kept boxes, at most max_det rows, and NMS of the kept set keeps all of it (idempotence)."""
    from efficientteacher_amd.utils.general import nms_padded, nms_ssod_padded
    B, A, nc = 32, 25200, 80
    g = torch.Generator(device="cpu").manual_seed(11)
    pred = torch.zeros(B, A, 5 + nc)
    pred[..., 0:2] = torch.rand(B, A, 2, generator=g) * 640
    pred[..., 2:4] = torch.rand(B, A, 2, generator=g) * 120 + 8
    pred[..., 4] = torch.rand(B, A, generator=g) ** 8
    pred[..., 5:] = torch.rand(B, A, nc, generator=g) ** 4
    pred = pred.to(dev)
    ct, it = (0.001, 0.65) if val_path else (0.1, 0.65)
    if val_path:
        dets, counts, _, _ = nms_padded(pred, ct, it, multi_label=True)
    else:
        dets, counts, _, _ = nms_ssod_padded(pred, ct, it)
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    assert (counts <= 300).all() and counts.max() > 0
    for i in range(B):
        d = dets[i, :counts[i]]
        assert (np.diff(d[:, 4]) <= 0).all() and (d[:, 4] > ct).all()
        off = d[:, 5:6] * 7680.0
        iou = _iou(d[:, :4] + off, d[:, :4] + off)
        np.fill_diagonal(iou, 0)
        assert iou.max() <= it + 1e-6
        assert (dets[i, counts[i]:] == 0).all()
    # idempotence on image 0: feed the kept boxes back as a one-class-per-row prediction
    d = dets[0, :counts[0]]
    p2 = np.zeros((1, len(d), 5 + nc), np.float32)
    p2[0, :, 0] = (d[:, 0] + d[:, 2]) / 2; p2[0, :, 1] = (d[:, 1] + d[:, 3]) / 2
    p2[0, :, 2] = d[:, 2] - d[:, 0]; p2[0, :, 3] = d[:, 3] - d[:, 1]
    p2[0, :, 4] = 1.0
    p2[0, np.arange(len(d)), 5 + d[:, 5].astype(int)] = d[:, 4]
    d2, c2, _, _ = nms_padded(torch.from_numpy(p2).to(dev), ct, it, multi_label=False)
    assert int(c2[0]) == len(d)


def test_ema_sgd_closed_forms_full_arena(dev):
    """flat-arena EMA / SGD on 48 M floats: EMA towards itself is the identity, EMA matches d*v + (1-d)*m, and
    SGD-nesterov matches the closed form of torch.optim.SGD for one step."""
    from efficientteacher_amd import ops
    n = 48_000_000
    g = torch.Generator(device="cpu").manual_seed(3)
    v = torch.randn(n, generator=g).to(dev)
    m = torch.randn(n, generator=g).to(dev)
    v0 = v.clone()
    ops.ema_update(v, v0, 0.9)
    assert torch.allclose(v, v0, rtol=0, atol=1e-6)     # 0.9 v + (1 - 0.9) v, one fp32 rounding per operation
    ops.ema_update(v, m, 0.9997)
    ref = 0.9997 * v0 + (1 - 0.9997) * m
    assert torch.allclose(v, ref, rtol=1e-6, atol=1e-6)
    p, grad, buf = m.clone(), torch.randn(n, generator=g).to(dev), torch.zeros(n, device=dev)
    ops.sgd_nesterov(p, grad, buf, None, 0.01, 0.937, 5e-4, True, 1.0)
    d_p = grad + 5e-4 * m
    ref_p = m - 0.01 * (d_p + 0.937 * d_p)              # first step: buf = d_p; nesterov: d_p + momentum * buf
    assert torch.allclose(p, ref_p, rtol=1e-5, atol=1e-6)
    assert torch.allclose(buf, d_p, rtol=1e-6, atol=1e-7)


def test_pseudo_label_round_trip_full_size(dev):
    """32 x 300 detections, identity warp: the pseudo labels are the detections in normalised xywh (to fp32
    rounding: the reference does the box conversions in fp32 before its fp64 warp); with the lr / ud flips set, x -> 1 - x and y -> 1 - y."""
    from efficientteacher_amd import ops
    B, D, W, H = 32, 300, 640, 640
    g = torch.Generator(device="cpu").manual_seed(21)
    cx = torch.rand(B, D, generator=g) * 400 + 120
    cy = torch.rand(B, D, generator=g) * 400 + 120
    w = torch.rand(B, D, generator=g) * 150 + 10
    h = torch.rand(B, D, generator=g) * 150 + 10
    dets = torch.zeros(B, D, 8)
    dets[..., 0], dets[..., 1], dets[..., 2], dets[..., 3] = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
    dets[..., 4] = torch.rand(B, D, generator=g)
    dets[..., 5] = torch.randint(0, 80, (B, D), generator=g).float()
    dets[..., 6] = torch.rand(B, D, generator=g)
    dets[..., 7] = torch.rand(B, D, generator=g)
    counts = torch.randint(1, D + 1, (B,), generator=g).int()
    M_s = torch.zeros(B, 13, dtype=torch.float64)
    M_s[:, 0] = torch.arange(B)
    M_s[:, 1], M_s[:, 5], M_s[:, 9], M_s[:, 10] = 1, 1, 1, 1          # M = I, s = 1
    d = dets.to(dev)
    t9, valid = ops.pseudo_label_transform(d, counts.to(dev), M_s.to(dev), W, H)
    t9, valid = t9.cpu().view(B, D, 9), valid.cpu().view(B, D).bool()
    for i in range(B):
        n = int(counts[i])
        assert valid[i, :n].all() and not valid[i, n:].any()
    dd = dets.double()
    x1, y1, x2, y2 = dd[..., 0], dd[..., 1], dd[..., 2], dd[..., 3]
    ref = torch.stack(((x1 + x2) / 2 / W, (y1 + y2) / 2 / H, (x2 - x1) / W, (y2 - y1) / H), -1)
    # the reference converts xyxy -> xywh -> corners in fp32 before the fp64 warp: agreement is fp32-exact, not fp64
    assert torch.allclose(t9[..., 2:6][valid], ref[valid], rtol=0, atol=2e-7)
    assert torch.equal(t9[..., 1][valid], dd[..., 5][valid]) and torch.equal(t9[..., 0][valid].long(), torch.arange(B)[:, None].expand(B, D)[valid])
    M_f = M_s.clone(); M_f[:, 11] = 1; M_f[:, 12] = 1
    t9f, vf = ops.pseudo_label_transform(d, counts.to(dev), M_f.to(dev), W, H)
    t9f = t9f.cpu().view(B, D, 9)
    assert torch.allclose(t9f[..., 2][valid], 1 - t9[..., 2][valid], rtol=0, atol=1e-12)
    assert torch.allclose(t9f[..., 3][valid], 1 - t9[..., 3][valid], rtol=0, atol=1e-12)
    assert torch.equal(t9f[..., 4:6][valid], t9[..., 4:6][valid])


def test_yolov5s_full_resolution_step_vs_oracle(dev):
    """BASELINE configs[0] scale: YOLOv5s widths (0.50 / 0.33), 2 x 3 x 640 x 640, fp32 parity mode: train
    forward + ComputeLoss + backward on the HIP kernels vs the plain-torch oracle model on the same weights."""
    import os
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from efficientteacher_amd.models.loss import ComputeLoss
    from oracle import losses as o_loss, model as o_model
    from tests.conftest import ROOT
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "efficientteacher_amd", "configs", "ssod", "coco-standard",
                                     "yolov5l_coco_ssod_10_percent.yaml"))
    cfg.merge_from_list(["Model.width_multiple", 0.50, "Model.depth_multiple", 0.33])
    torch.manual_seed(0)
    model = Model(cfg)
    ref = o_model.Model.from_cfg(cfg)
    ref.load_state_dict(model.state_dict(), strict=True)
    model = model.to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.rand(2, 3, 640, 640, generator=g)
    targets = torch.tensor([[0, 3, .5, .5, .2, .3], [1, 17, .3, .6, .1, .1], [1, 0, .7, .2, .4, .5]])
    closs = ComputeLoss(model, cfg)
    pred, _ = model(x.to(dev))
    loss, items = closs(pred, targets.to(dev))
    loss.backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rp, _ = ref.train()(x)
    rl, _ = o_loss.compute_loss(rp, targets, ref.head.anchors, nc=80, box_w=closs.box_w, obj_w=closs.obj_w,
                                cls_w=closs.cls_w)
    rl.backward()
    for a, b in zip(pred, rp):
        assert (a.detach().cpu() - b.detach()).abs().max().item() < 1e-3
    assert abs(loss.item() - rl.item()) <= 1e-4 * abs(rl.item()), (loss.item(), rl.item())
    gp, gr = dict(model.named_parameters()), dict(ref.named_parameters())
    for name in ("backbone.stage1.conv.weight", "backbone.stage3_2.cv3.conv.weight", "neck.C3.m.0.cv2.conv.weight",
                 "head.m.1.weight", "backbone.stage2_2.cv1.bn.weight"):
        a, b = gp[name].grad.cpu(), gr[name].grad
        assert (a - b).abs().max().item() <= 5e-3 * max(b.abs().max().item(), 1e-6), name
