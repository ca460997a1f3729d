"""Fused assignment + loss kernels vs the reference's golden outputs (ComputeLoss,
ComputeStudentMatchLoss incl. select_targets) and vs the oracle on fresh seeds; pseudo-label kernel."""
import types

import numpy as np
import pytest
import torch

from oracle import losses as o_loss
from oracle import nms as o_nms
from oracle import pseudo_label as o_pl
from tests.conftest import golden


def _cfg():
    import os
    from efficientteacher_amd.configs import get_cfg
    from tests.conftest import ROOT
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "efficientteacher_amd/configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml"))
    return cfg


def _fake_model(anchors, dev):
    head = types.SimpleNamespace(nl=3, na=3, nc=80, num_keypoints=0, anchors=torch.as_tensor(anchors).to(dev),
                                 stride=torch.tensor([8., 16., 32.]))
    return types.SimpleNamespace(head=head)


def test_compute_loss_golden(hip):
    from efficientteacher_amd.models.loss import ComputeLoss
    g = golden("compute_loss")
    closs = ComputeLoss(_fake_model(g["anchors"], hip.device), _cfg())
    assert np.allclose([closs.box_w, closs.obj_w, closs.cls_w, closs.anchor_t], g["weights"])
    p = [hip.t(g[f"p{i}"]).requires_grad_(True) for i in range(3)]
    loss, items = closs(p, hip.t(g["targets"]))
    assert abs(loss.item() - float(g["loss"][0])) <= 1e-4 * abs(float(g["loss"][0]))
    assert np.allclose([items[k].item() for k in ("box", "obj", "cls")], g["items"], rtol=1e-4, atol=1e-6)
    loss.backward()
    for i in range(3):
        ref = g[f"grad{i}"]
        err = np.abs(p[i].grad.cpu().numpy() - ref).max()
        assert err <= 1e-4 * np.abs(ref).max() + 1e-7, (i, err)
    l0, _ = closs([hip.t(g[f"p{i}"]) for i in range(3)], torch.zeros((0, 6), device=hip.device))
    assert abs(l0.item() - float(g["loss_empty"][0])) <= 1e-4 * abs(float(g["loss_empty"][0]))


@pytest.mark.parametrize("tag", ["default", "cls", "ignore"])
def test_student_match_loss_golden(hip, tag):
    from efficientteacher_amd.models.loss import ComputeStudentMatchLoss
    g = golden("student_match_loss")
    gl = golden("compute_loss")
    s = ComputeStudentMatchLoss(_fake_model(g["anchors"], hip.device), _cfg())
    if tag == "cls":
        s.pseudo_label_with_cls = True
    if tag == "ignore":
        s.ignore_obj = True
    p = [hip.t(gl[f"p{i}"]).requires_grad_(True) for i in range(3)]
    loss, items = s(p, hip.t(g["targets9"]))
    ref = float(g[f"{tag}_loss"][0])
    assert abs(loss.item() - ref) <= 1e-4 * abs(ref), (loss.item(), ref)
    assert np.allclose([items[k].item() for k in ("ss_box", "ss_obj", "ss_cls")], g[f"{tag}_items"], rtol=1e-4, atol=1e-6)
    loss.backward()
    for i in range(3):
        r = g[f"{tag}_grad{i}"]
        assert np.abs(p[i].grad.cpu().numpy() - r).max() <= 1e-4 * np.abs(r).max() + 1e-7
    table = s.select_targets(hip.t(g["targets9"])).cpu().numpy()
    flags = table[:, 7].astype(int)
    counts = [(flags & 1).sum(), ((flags >> 1) & 1).sum(), ((flags >> 2) & 1).sum(), ((flags >> 3) & 1).sum()]
    assert counts == list(g["sel_counts"])


@pytest.mark.parametrize("seed", [0, 1])
def test_loss_vs_oracle_with_duplicates_and_views(hip, seed):
    """Many targets on a small grid (duplicate cells: last-writer-wins rule), logits as strided
    head views over a (B, ny, nx, 256) buffer, bf16 logits."""
    from efficientteacher_amd.autograd import head_view
    from efficientteacher_amd.models.loss import ComputeLoss
    rng = np.random.default_rng(seed)
    g = golden("compute_loss")
    B = 2
    shapes = [(8, 8), (4, 4), (2, 2)]
    nt = 60
    t = np.zeros((nt, 6), np.float32)
    t[:, 0] = np.sort(rng.integers(0, B, nt)); t[:, 1] = rng.integers(0, 80, nt)
    t[:, 2:4] = rng.uniform(0.05, 0.95, (nt, 2)); t[:, 4:6] = np.exp(rng.uniform(np.log(0.05), np.log(0.8), (nt, 2)))
    bufs = [torch.from_numpy(rng.normal(0, 1.2, (B, ny, nx, 256)).astype(np.float32)) for ny, nx in shapes]
    closs = ComputeLoss(_fake_model(g["anchors"], hip.device), _cfg())
    pv = [head_view(hip.t(b), 3, 85).requires_grad_(True) for b in bufs]
    loss, _ = closs(pv, hip.t(t))
    pr = [head_view(b.clone(), 3, 85).contiguous().requires_grad_(True) for b in bufs]
    lref, _ = o_loss.compute_loss(pr, torch.from_numpy(t), torch.from_numpy(g["anchors"]), nc=80, box_w=closs.box_w,
                                  obj_w=closs.obj_w, cls_w=closs.cls_w, anchor_t=closs.anchor_t)
    assert abs(loss.item() - lref.item()) <= 1e-4 * abs(lref.item())
    loss.backward(); lref.backward()
    for a, b in zip(pv, pr):
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 1e-4 * b.grad.abs().max().item() + 1e-7


def test_pseudo_label_golden(hip):
    from efficientteacher_amd import ops
    from efficientteacher_amd.utils.general import nms_ssod_padded
    g = golden("pseudo_label")
    H, W = g["hw"]
    dets, counts, _, _ = nms_ssod_padded(hip.t(g["pred"]), float(g["thr"][0]), float(g["thr"][1]))
    t9, valid = ops.pseudo_label_transform(dets, counts, hip.t(g["M_s"]), W, H)
    got = t9[valid.bool()].cpu().numpy()
    assert got.shape == g["targets"].shape
    assert np.allclose(got, g["targets"], rtol=1e-12, atol=1e-12)
    # oracle on a second configuration (non-identity M on every image, flips)
    rng = np.random.default_rng(3)
    pred = g["pred"].copy(); pred[..., :2] += rng.normal(0, 3, pred[..., :2].shape).astype(np.float32)
    M_s = g["M_s"].copy(); M_s[:, 3] += 11.5; M_s[:, 11] = 1; M_s[:, 12] = [0, 1, 1]
    d2, c2, _, _ = nms_ssod_padded(hip.t(pred), 0.1, 0.65)
    t9, valid = ops.pseudo_label_transform(d2, c2, hip.t(M_s), W, H)
    refd, _ = o_nms.non_max_suppression_ssod(pred, 0.1, 0.65)
    reft, inv = o_pl.create_pseudo_label(refd, M_s, W, H)
    assert np.allclose(t9[valid.bool()].cpu().numpy(), reft, rtol=1e-12, atol=1e-12)


def test_domain_and_target_loss_golden(hip):
    """DomainLoss / TargetLoss (softmax focal, models/loss/loss.py:312-421) forward + gradient."""
    from efficientteacher_amd.models.loss import DomainLoss, TargetLoss
    g = golden("domain_loss")
    feats = []
    for i in range(3):
        f = g[f"f{i}"]                                   # (B,2,H,W)
        buf = torch.zeros((f.shape[0], f.shape[2], f.shape[3], 8), dtype=torch.float32)
        buf[..., :2] = torch.from_numpy(f).permute(0, 2, 3, 1)
        feats.append(hip.t(buf).requires_grad_(True))
    views = [b[..., :2].permute(0, 3, 1, 2) for b in feats]     # what netD.forward returns
    d = DomainLoss()(views)
    t = TargetLoss()(views)
    assert abs(d.item() - float(g["d"])) <= 1e-5 * abs(float(g["d"])) + 1e-7
    assert abs(t.item() - float(g["t"])) <= 1e-5 * abs(float(g["t"])) + 1e-7
    (d + 2 * t).backward()
    for i in range(3):
        ref = torch.from_numpy(g[f"g{i}"]).permute(0, 2, 3, 1)
        got = feats[i].grad[..., :2].cpu()
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-9
        assert (feats[i].grad[..., 2:] == 0).all()


def test_focal_compute_loss_golden(hip):
    """Loss.fl_gamma = 1.5, label smoothing 0.1: FocalLoss around the class and objectness BCE (reference loss.py:37-62),
    fused value + gradient against the reference run (tests/golden/focal_loss.npz, `python -m oracle.make_golden focal`)"""
    from efficientteacher_amd.models.loss import ComputeLoss
    g, gi = golden("focal_loss"), golden("compute_loss")
    cfg = _cfg()
    cfg.merge_from_list(["Loss.fl_gamma", float(g["hp"][0]), "Loss.label_smoothing", float(g["hp"][1])])
    closs = ComputeLoss(_fake_model(gi["anchors"], hip.device), cfg)
    assert closs.fl_gamma == 1.5 and abs(closs.cp - g["hp"][2]) < 1e-12 and abs(closs.cn - g["hp"][3]) < 1e-12
    p = [hip.t(gi[f"p{i}"]).requires_grad_(True) for i in range(3)]
    loss, items = closs(p, hip.t(gi["targets"]))
    ref = float(g["loss"][0])
    assert abs(loss.item() - ref) <= 1e-4 * abs(ref), (loss.item(), ref)
    assert np.allclose([items[k].item() for k in ("box", "obj", "cls")], g["items"], rtol=1e-4, atol=1e-7)
    loss.backward()
    for i in range(3):
        r = g[f"grad{i}"]
        assert np.abs(p[i].grad.cpu().numpy() - r).max() <= 1e-4 * np.abs(r).max() + 1e-8, i


def test_autobalance_golden(hip):
    """Loss.autobalance: three consecutive calls; the objectness balance weights are updated on the device after every call
    exactly as the reference updates its python list (loss.py:193-197) -- losses, weights and the level-0 objectness gradient"""
    from efficientteacher_amd.models.loss import ComputeLoss
    g, gi = golden("autobalance"), golden("compute_loss")
    cfg = _cfg()
    cfg.merge_from_list(["Loss.autobalance", True])
    closs = ComputeLoss(_fake_model(gi["anchors"], hip.device), cfg)
    assert closs.autobalance and closs.ssi == 1
    t = hip.t(gi["targets"])
    for k in range(3):
        p = [(hip.t(gi[f"p{i}"]) * (1.0 + 0.25 * k)).requires_grad_(True) for i in range(3)]
        loss, _ = closs(p, t)
        loss.backward()
        ref = float(g[f"loss{k}"][0])
        assert abs(loss.item() - ref) <= 1e-4 * abs(ref), (k, loss.item(), ref)
        assert np.allclose(closs._balance_dev.cpu().numpy(), g[f"balance{k}"], rtol=1e-5), k
        r = g[f"grad{k}"]
        assert np.abs(p[0].grad[..., 4].cpu().numpy() - r).max() <= 1e-4 * np.abs(r).max() + 1e-9, k
