"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (read-only /root/reference).

TEST INFRASTRUCTURE; build-container only (``python -m oracle.make_golden``).
Every case stores the seeded inputs and the outputs of the reference's own
functions.  While generating, the oracle restatements in ``oracle/`` are checked
against the reference outputs (the "pin"); ``tests/test_oracle_golden.py`` repeats
that check from the committed files on every run, with no reference present.

Reference entry points exercised (file:line):
  utils/general.py:887 non_max_suppression_ssod, :994 non_max_suppression
  models/assigner/yolo_anchor_assigner.py:319 build_targets, :640 build_uc_targets_aug
  utils/metrics.py:207 bbox_iou(CIoU)
  models/loss/loss.py:93 ComputeLoss (:138 default_loss, :210 ota_loss), :376 TargetLoss, :398 DomainLoss
  models/assigner/yolo_anchor_assigner.py:104 build_ota_targets, :266 find_3_positive
  models/loss/ssod/ssod_loss.py:26 ComputeStudentMatchLoss
  utils/self_supervised_utils.py:194 FairPseudoLabel.create_pseudo_label_online_with_gt
  utils/labelmatch.py:56 LabelMatch (:271 create_pseudo_label_online_with_gt, :188 update_epoch_cls_thr)
  models/detector/yolo_ssod.py:44 Model (train + eval forward, backward)
  utils/torch_utils.py:308 ModelEMA, :381 CosineEMA ; torch.optim.SGD as trainer.py:215
"""
import copy
import os
import sys

import numpy as np
import torch

from . import assigner as o_asg
from . import detect as o_det
from . import losses as o_loss
from . import nms as o_nms
from . import optim as o_opt
from . import pseudo_label as o_pl
from . import ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SSOD_YAML = "configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml"
TINY = ["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33]


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def synth_pred(rng, B, A, nc, obj_pow=4, cls_pow=4, dup=False):
    """Decoded-prediction-like tensor (B,A,5+nc): xywh px, obj, cls."""
    p = np.zeros((B, A, 5 + nc), np.float32)
    p[..., 0:2] = rng.uniform(0, 640, (B, A, 2))
    p[..., 2:4] = rng.uniform(4, 220, (B, A, 2))
    p[..., 4] = rng.uniform(0, 1, (B, A)) ** obj_pow
    p[..., 5:] = rng.uniform(0, 1, (B, A, nc)) ** cls_pow
    if dup:  # exact score ties + identical boxes exercise the stable-sort rule
        p[:, 1::7] = p[:, 0:-1:7][:, : p[:, 1::7].shape[1]]
    return p


def case_nms():
    from utils.general import non_max_suppression, non_max_suppression_ssod
    rng = np.random.default_rng(11)
    cases = {
        "a": (synth_pred(rng, 2, 900, 20), 0.1, 0.65),
        "b": (synth_pred(rng, 3, 700, 8, obj_pow=1, cls_pow=1, dup=True), 0.1, 0.65),
        "c": (synth_pred(rng, 2, 300, 5, obj_pow=8), 0.25, 0.45),
        "empty": (synth_pred(rng, 2, 64, 4) * np.float32(0.05), 0.1, 0.65),
    }
    # (ORACLE-DERIVED case: with torchvision absent, the reference's torchvision.ops.nms below IS oracle.nms.nms_torch -- the
    # `ref == mine` assertion of this function then compares the restatement with itself for the tie rule; oracle/nms.py header)
    # exact tie at the threshold: IoU == fp32(0.6) -- xyxy [100,100,110,110] (area 100) against [100,100,106,110] (inside it, area 60):
    # inter / union = 60 / 100, and the correctly rounded fp32 quotient IS fp32(0.6).  The pinned compare (torchvision's CUDA kernel,
    # fp32 threshold: oracle/nms.py header) KEEPS the second box, the CPU kernel's double compare would drop it; rows 2-3 / 4-5 are
    # the controls one ulp-ish either side (6.01 / 5.99 wide), 6-7 an unrelated pair
    tie = np.zeros((1, 8, 5 + 2), np.float32)
    tie[0, :, 0:4] = [[105, 105, 10, 10], [103, 105, 6, 10], [305, 105, 10, 10], [303.005, 105, 6.01, 10],
                      [505, 105, 10, 10], [502.995, 105, 5.99, 10], [105, 305, 10, 10], [400, 400, 30, 30]]
    tie[0, :, 4] = [0.9, 0.8, 0.9, 0.8, 0.9, 0.8, 0.7, 0.6]
    tie[0, :, 5] = 0.95
    cases["tie06"] = (tie, 0.1, 0.6)
    bx = o_nms.xywh2xyxy(tie[0, :2, :4])
    assert o_nms.nms(bx, tie[0, :2, 4], 0.6).tolist() == [0, 1] and o_nms.nms(bx, tie[0, :2, 4], 0.6, "cpu_double").tolist() == [0], \
        "tie06 must separate the fp32 (CUDA) threshold compare from the double (CPU) one"
    # cluster boxes so that suppression actually happens
    for k in ("a", "b"):
        p = cases[k][0]
        centers = rng.uniform(100, 540, (p.shape[0], 12, 2)).astype(np.float32)
        idx = rng.integers(0, 12, p.shape[:2])
        p[..., 0:2] = np.take_along_axis(centers, idx[..., None].repeat(2, 2), 1) + \
            rng.normal(0, 6, p.shape[:2] + (2,)).astype(np.float32)
        p[..., 2:4] = 80 + rng.normal(0, 8, p.shape[:2] + (2,)).astype(np.float32)
    out = {}
    for k, (pred, ct, it) in cases.items():
        ref = non_max_suppression_ssod(torch.from_numpy(pred.copy()), ct, it)
        mine, keeps = o_nms.non_max_suppression_ssod(pred, ct, it)
        for r, m in zip(ref, mine):
            r = r.numpy().reshape(-1, 8) if r.shape[-1] == 8 else np.zeros((0, 8), np.float32)
            assert np.array_equal(r, m), f"nms_ssod pin failed ({k})"
        out[f"{k}_pred"] = pred
        out[f"{k}_thr"] = np.array([ct, it], np.float64)
        out[f"{k}_counts"] = np.array([m.shape[0] for m in mine], np.int64)
        out[f"{k}_dets"] = np.concatenate(mine, 0)
        out[f"{k}_keep"] = np.concatenate(keeps, 0)
        # val-path NMS (multi_label) on the same tensors  (row f-1)
        refv = non_max_suppression(torch.from_numpy(pred.copy()), ct, it, multi_label=True)
        minev = o_nms.non_max_suppression(pred, ct, it, multi_label=True)
        for r, m in zip(refv, minev):
            assert np.array_equal(r.numpy().reshape(-1, 6), m), f"nms(val) pin failed ({k})"
        out[f"{k}_val_counts"] = np.array([m.shape[0] for m in minev], np.int64)
        out[f"{k}_val_dets"] = np.concatenate(minev, 0)
    save("nms", **out)


def case_nms_ssod_options():
    """non_max_suppression_ssod's optional arguments (utils/general.py:887: classes, multi_label, labels, agnostic) -- off every
    shipped SSOD config, pinned so that the drop-in signature is complete"""
    from utils.general import non_max_suppression_ssod
    rng = np.random.default_rng(23)
    pred = synth_pred(rng, 3, 500, 6, obj_pow=1, cls_pow=1, dup=True)
    centers = rng.uniform(100, 540, (3, 10, 2)).astype(np.float32)
    idx = rng.integers(0, 10, pred.shape[:2])
    pred[..., 0:2] = np.take_along_axis(centers, idx[..., None].repeat(2, 2), 1) + rng.normal(0, 6, pred.shape[:2] + (2,)).astype(np.float32)
    pred[..., 2:4] = 80 + rng.normal(0, 8, pred.shape[:2] + (2,)).astype(np.float32)
    labels = [np.array([[2, 300, 300, 90, 90], [5, 120, 140, 60, 70]], np.float32), np.zeros((0, 5), np.float32),
              np.array([[0, 500, 480, 100, 80]], np.float32)]
    variants = {
        "classes": dict(classes=[1, 4]),
        "classes_agnostic": dict(classes=[0, 2, 5], agnostic=True),
        "multi_label": dict(multi_label=True),
        "multi_label_classes": dict(multi_label=True, classes=[3]),
        "labels": dict(labels=[torch.from_numpy(l) for l in labels]),
        "labels_multi": dict(labels=[torch.from_numpy(l) for l in labels], multi_label=True, agnostic=True),
    }
    out = dict(pred=pred, thr=np.array([0.1, 0.6], np.float64), apriori_rows=np.concatenate(labels, 0),
               apriori_counts=np.array([len(l) for l in labels], np.int64))
    for k, kw in variants.items():
        ref = non_max_suppression_ssod(torch.from_numpy(pred.copy()), 0.1, 0.6, **kw)
        okw = dict(kw)
        if "labels" in okw:
            okw["labels"] = labels
        mine, keeps = o_nms.non_max_suppression_ssod(pred, 0.1, 0.6, **okw)
        w = 6 if kw.get("multi_label") else 8
        for r, m in zip(ref, mine):
            r = r.numpy().reshape(-1, w) if r.shape[0] else np.zeros((0, w), np.float32)
            assert np.array_equal(r, m), f"nms_ssod options pin failed ({k})"
        assert sum(m.shape[0] for m in mine) > 0, k
        out[f"{k}_counts"] = np.array([m.shape[0] for m in mine], np.int64)
        out[f"{k}_dets"] = np.concatenate(mine, 0)
    save("nms_ssod_options", **out)


def synth_targets(rng, B, n_per=(1, 9), with_edge=True):
    rows = []
    for b in range(B):
        n = int(rng.integers(*n_per))
        xy = rng.uniform(0.02, 0.98, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.02), np.log(0.7), (n, 2)))
        cls = rng.integers(0, 80, (n, 1))
        rows.append(np.concatenate((np.full((n, 1), b), cls, xy, wh), 1))
    t = np.concatenate(rows, 0).astype(np.float32)
    if with_edge:  # cells on the image border / exact half-cell positions
        t[0, 2:4] = [0.999, 0.0005]
        t[-1, 2:4] = [0.5, 0.5]
    return t


def build_tiny_model(nc=80, seed=0):
    cfg = ref_loader.get_cfg(SSOD_YAML, TINY + ["Dataset.nc", nc])
    cfg.freeze()
    from models.detector.yolo_ssod import Model
    torch.manual_seed(seed)
    model = Model(cfg)
    # default init leaves BN weight=1,bias=0 and running stats trivial: perturb so that the
    # eval path (running stats) and affine terms are exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_var.copy_(1 + 0.3 * torch.rand(m.bias.shape, generator=g))
    return cfg, model


def case_assigner_and_losses():
    cfg, model = build_tiny_model()
    from models.loss.loss import ComputeLoss, DomainLoss, TargetLoss
    from models.loss.ssod.ssod_loss import ComputeStudentMatchLoss
    from utils.metrics import bbox_iou
    det = model.head
    anchors = det.anchors.clone()
    rng = np.random.default_rng(5)
    B = 3
    shapes = [(16, 16), (8, 8), (4, 4)]        # a 128x128 image
    targets = synth_targets(rng, B)
    p = [torch.from_numpy(rng.normal(0, 1.5, (B, 3, ny, nx, 85)).astype(np.float32)) for ny, nx in shapes]

    # --- build_targets / build_uc_targets_aug ---------------------------------
    closs = ComputeLoss(model, cfg)
    ref = closs.assigner(p, torch.from_numpy(targets))
    mine = o_asg.build_targets(shapes, anchors.numpy(), targets, cfg.Loss.anchor_t)
    out = dict(targets=targets, anchors=anchors.numpy(), shapes=np.array(shapes),
               anchor_t=np.float64(cfg.Loss.anchor_t))
    for i in range(3):
        tcls, tbox, ind, anch = ref[0][i], ref[1][i], ref[2][i], ref[3][i]
        m = mine[i]
        assert np.array_equal(tcls.numpy(), m["tcls"]) and np.array_equal(tbox.numpy(), m["tbox"])
        assert all(np.array_equal(ind[k].numpy(), m[n]) for k, n in enumerate(("b", "a", "gj", "gi")))
        assert np.array_equal(anch.numpy(), m["anch"])
        for k, v in m.items():
            out[f"bt{i}_{k}"] = v
    t7 = np.concatenate((targets, rng.uniform(0.1, 1, (targets.shape[0], 1)).astype(np.float32)), 1)
    refu = closs.assigner(p, torch.from_numpy(t7), with_pseudo_score=True)
    mineu = o_asg.build_targets(shapes, anchors.numpy(), t7, cfg.Loss.anchor_t, with_score=True)
    for i in range(3):
        assert np.array_equal(refu[4][i].numpy(), mineu[i]["tscore"])
        assert np.array_equal(refu[2][i][3].numpy(), mineu[i]["gi"])
        out[f"uc{i}_tscore"] = mineu[i]["tscore"]
        out[f"uc{i}_gi"] = mineu[i]["gi"]
    out["targets7"] = t7
    save("assigner", **out)

    # --- CIoU ------------------------------------------------------------------
    n = 257
    pb = np.abs(rng.normal(1.0, 0.8, (n, 4))).astype(np.float32) + 0.01
    tb = np.abs(rng.normal(1.0, 0.8, (n, 4))).astype(np.float32) + 0.01
    pbt = torch.from_numpy(pb).requires_grad_(True)
    iou = bbox_iou(pbt.T, torch.from_numpy(tb), x1y1x2y2=False, CIoU=True)
    (1 - iou).mean().backward()
    pbt2 = torch.from_numpy(pb).requires_grad_(True)
    iou2 = o_loss.ciou_xywh(pbt2, torch.from_numpy(tb))
    (1 - iou2).mean().backward()
    assert torch.equal(iou, iou2) and torch.equal(pbt.grad, pbt2.grad), "ciou pin failed"
    save("ciou", pbox=pb, tbox=tb, iou=iou.detach().numpy(), grad=pbt.grad.numpy())

    # --- ComputeLoss -------------------------------------------------------------
    out = dict(targets=targets, anchors=anchors.numpy())
    pr = [x.clone().requires_grad_(True) for x in p]
    loss, items = closs(pr, torch.from_numpy(targets))
    loss.backward()
    kw = dict(nc=80, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w, anchor_t=closs.anchor_t)
    po = [x.clone().requires_grad_(True) for x in p]
    loss2, items2 = o_loss.compute_loss(po, torch.from_numpy(targets), anchors, **kw)
    loss2.backward()
    assert torch.allclose(loss, loss2, rtol=1e-6, atol=1e-7), (loss, loss2)
    for a, b in zip(pr, po):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8)
    for i in range(3):
        out[f"p{i}"] = p[i].numpy()
        out[f"grad{i}"] = pr[i].grad.numpy()
    out["loss"] = loss.detach().numpy()
    out["items"] = np.array([items[k].item() for k in ("box", "obj", "cls")], np.float32)
    out["weights"] = np.array([closs.box_w, closs.obj_w, closs.cls_w, closs.anchor_t], np.float64)
    # empty-target case (loss.py:149 n == 0 branch)
    l0, _ = closs([x.clone() for x in p], torch.zeros((0, 6)))
    l0o, _ = o_loss.compute_loss([x.clone() for x in p], torch.zeros((0, 6)), anchors, **kw)
    assert torch.allclose(l0, l0o, rtol=1e-6)
    out["loss_empty"] = l0.detach().numpy()
    save("compute_loss", **out)

    # --- ComputeStudentMatchLoss -------------------------------------------------
    sloss = ComputeStudentMatchLoss(model, cfg)
    nt = 40
    t9 = np.zeros((nt, 9), np.float64)
    t9[:, 0] = rng.integers(0, B, nt)
    t9[:, 0].sort()
    t9[:, 1] = rng.integers(0, 80, nt)
    t9[:, 2:4] = rng.uniform(0.05, 0.95, (nt, 2))
    t9[:, 4:6] = np.exp(rng.uniform(np.log(0.03), np.log(0.6), (nt, 2)))
    t9[:, 7] = rng.uniform(0.1, 1, nt) ** 0.3          # obj conf
    t9[:, 8] = rng.uniform(0.1, 1, nt) ** 0.3          # cls conf
    t9[:, 6] = t9[:, 7] * t9[:, 8]
    t9[::5, 7] = 0.995                                  # force the uc_obj branch
    t9[1::5, 8] = 0.999
    t9[3, 6] = 0.6                                      # exactly on the high threshold
    t9[4, 6] = 0.1                                      # exactly on the low threshold
    t9[5, 6] = 0.05                                     # below low: dropped
    out = dict(targets9=t9, anchors=anchors.numpy())
    for tag, flags in (("default", {}), ("cls", dict(pseudo_label_with_cls=True)),
                       ("ignore", dict(ignore_obj=True))):
        s = copy.copy(sloss)
        for k, v in flags.items():
            setattr(s, k, v)
        pr = [x.clone().requires_grad_(True) for x in p]
        loss, items = s(pr, torch.from_numpy(t9))
        loss.backward()
        po = [x.clone().requires_grad_(True) for x in p]
        loss2, _ = o_loss.compute_student_match_loss(
            po, torch.from_numpy(t9), anchors, nc=80, box_w=s.box_w, obj_w=s.obj_w, cls_w=s.cls_w,
            anchor_t=s.anchor_t, thr_low=s.ignore_thres_low, thr_high=s.ignore_thres_high,
            ignore_obj=s.ignore_obj, with_obj=s.pseudo_label_with_obj,
            with_bbox=s.pseudo_label_with_bbox, with_cls=s.pseudo_label_with_cls)
        loss2.backward()
        assert torch.allclose(loss, loss2, rtol=1e-6, atol=1e-7), (tag, loss, loss2)
        for a, b in zip(pr, po):
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8), tag
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_items"] = np.array([float(items[k]) for k in ("ss_box", "ss_obj", "ss_cls")], np.float32)
        for i in range(3):
            out[f"{tag}_grad{i}"] = pr[i].grad.numpy()
    rel, unc, uo, ucl = sloss.select_targets(torch.from_numpy(t9))
    mr = o_loss.select_targets(t9, sloss.ignore_thres_low, sloss.ignore_thres_high)
    for a, b in zip((rel, unc, uo, ucl), mr):
        assert np.array_equal(a.numpy().reshape(-1, 7), b)
    out["sel_counts"] = np.array([x.shape[0] for x in mr], np.int64)
    out["weights"] = np.array([sloss.box_w, sloss.obj_w, sloss.cls_w, sloss.anchor_t], np.float64)
    save("student_match_loss", **out)

    # --- Domain / Target loss --------------------------------------------------
    feats = [torch.from_numpy(rng.normal(0, 1, (2, 2, s, s)).astype(np.float32)) for s in (8, 4, 2)]
    fr = [f.clone().requires_grad_(True) for f in feats]
    d = DomainLoss()(fr); t = TargetLoss()(fr); (d + 2 * t).backward()
    fo = [f.clone().requires_grad_(True) for f in feats]
    d2 = o_loss.domain_loss(fo, 0); t2 = o_loss.domain_loss(fo, 1); (d2 + 2 * t2).backward()
    assert torch.allclose(d, d2, rtol=1e-6) and torch.allclose(t, t2, rtol=1e-6)
    for a, b in zip(fr, fo):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8)
    save("domain_loss", f0=feats[0].numpy(), f1=feats[1].numpy(), f2=feats[2].numpy(),
         d=d.detach().numpy(), t=t.detach().numpy(),
         g0=fr[0].grad.numpy(), g1=fr[1].grad.numpy(), g2=fr[2].grad.numpy())
    return cfg, model


def case_pseudo_label(cfg):
    from utils.self_supervised_utils import FairPseudoLabel
    rng = np.random.default_rng(21)
    B, A, nc = 3, 500, 80
    W = H = 640
    pred = synth_pred(rng, B, A, nc, obj_pow=2, cls_pow=6)
    pred[..., 5:] = 0
    hot = rng.integers(0, nc, (B, A))
    np.put_along_axis(pred[..., 5:], hot[..., None], rng.uniform(0.3, 1, (B, A, 1)).astype(np.float32), 2)
    M_s = np.zeros((B, 13), np.float64)
    for i in range(B):
        s = [1.0, 0.8, 1.25][i]
        M = np.array([[s, 0.02 * i, 30.0 * i - 20], [-0.03 * i, s, 12.0 * i], [0, 0, 1]], np.float64)
        M_s[i] = [i, *M.reshape(-1), s, i % 2, (i + 1) % 2]
    imgs = torch.zeros(B, 3, H, W)
    fpl = FairPseudoLabel(cfg)
    ref_t, ref_invalid = fpl.create_pseudo_label_online_with_gt(
        torch.from_numpy(pred.copy()), imgs, torch.from_numpy(M_s), imgs.clone())
    dets, _ = o_nms.non_max_suppression_ssod(pred, cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    mine_t, mine_invalid = o_pl.create_pseudo_label(dets, M_s, W, H)
    assert ref_invalid == mine_invalid and ref_t.dtype == torch.float64
    assert np.array_equal(ref_t.numpy(), mine_t), "pseudo-label pin failed"
    # no-detection case
    ref_e, inv_e = fpl.create_pseudo_label_online_with_gt(
        torch.from_numpy(pred * np.float32(0.01)), imgs, torch.from_numpy(M_s), imgs.clone())
    assert inv_e is True and len(ref_e) == 0
    save("pseudo_label", pred=pred, M_s=M_s, targets=mine_t, hw=np.array([H, W]),
         thr=np.array([cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres]))


def state_arrays(model):
    return {k.replace(".", "__"): v.detach().numpy() for k, v in model.state_dict().items()}


def case_model(cfg, model):
    """Tiny-width SSOD model (yolo_ssod.py:44): eval + train forward, loss backward."""
    from models.loss.loss import ComputeLoss
    rng = np.random.default_rng(33)
    B, S = 2, 64
    x = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    targets = synth_targets(rng, B, n_per=(2, 5), with_edge=False)
    sd = state_arrays(model)
    out = {f"w__{k}": v for k, v in sd.items()}
    out["x"] = x
    out["targets"] = targets
    out["stride"] = model.stride.numpy()
    out["anchors"] = model.head.anchors.numpy()
    # eval forward (teacher path, ssod_trainer.py:599)
    m = copy.deepcopy(model).eval()
    with torch.no_grad():
        (z, xs), feats = m(torch.from_numpy(x))
    zo = o_det.decode([t.clone() for t in xs], m.head.anchors, m.head.stride)
    assert torch.allclose(z, zo, rtol=1e-6, atol=1e-6), "decode pin failed"
    out["eval_z"] = z.numpy()
    for i in range(3):
        out[f"eval_x{i}"] = xs[i].numpy()
        out[f"eval_feat{i}"] = feats[i].numpy()
    # train forward + ComputeLoss backward (student path, ssod_trainer.py:626-628)
    m = copy.deepcopy(model).train()
    closs = ComputeLoss(m, cfg)
    pred, feats = m(torch.from_numpy(x))
    loss, items = closs(pred, torch.from_numpy(targets))
    loss.backward()
    out["train_loss"] = loss.detach().numpy()
    out["train_items"] = np.array([items[k].item() for k in ("box", "obj", "cls")], np.float32)
    for i in range(3):
        out[f"train_p{i}"] = pred[i].detach().numpy()
        out[f"train_feat{i}"] = feats[i].detach().numpy()
    for k, p in m.named_parameters():
        out["g__" + k.replace(".", "__")] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    for k, b in m.named_buffers():
        if "running" in k:
            out["b__" + k.replace(".", "__")] = b.numpy()
    save("model_tiny", **out)


def case_optim(model):
    from utils.torch_utils import CosineEMA, ModelEMA
    m = copy.deepcopy(model).train()
    # param groups exactly as trainer/trainer.py:199-223
    g0, g1, g2 = [], [], []
    for v in m.modules():
        if hasattr(v, "bias") and isinstance(v.bias, torch.nn.Parameter):
            g2.append(v.bias)
        if isinstance(v, torch.nn.BatchNorm2d):
            g0.append(v.weight)
        elif hasattr(v, "weight") and isinstance(v.weight, torch.nn.Parameter):
            g1.append(v.weight)
    opt = torch.optim.SGD(g0, lr=0.01, momentum=0.937, nesterov=True)
    opt.add_param_group({"params": g1, "weight_decay": 5e-4})
    opt.add_param_group({"params": g2})
    ema = ModelEMA(m)
    semi = CosineEMA(ema.ema, decay_start=0.999, decay_end=0.9999, total_epoch=300)
    gen = torch.Generator().manual_seed(9)
    p0 = {k: v.detach().clone() for k, v in m.named_parameters()}
    grads = []
    for step in range(2):
        gs = {}
        for k, p in m.named_parameters():
            p.grad = 0.01 * torch.randn(p.shape, generator=gen)
            gs[k] = p.grad.clone()
        grads.append(gs)
        opt.step()
        ema.update(m)
        semi.update(ema.ema)
    # oracle restatement
    pw = {k: v.numpy().copy() for k, v in p0.items()}
    wd = {id(p): 5e-4 for p in g1}
    buf = {}
    ema_o = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    semi_o = copy.deepcopy(ema_o)
    name_of = {id(p): k for k, p in m.named_parameters()}
    for step in range(2):
        for k, p in m.named_parameters():
            pw[k], buf[k] = o_opt.sgd_nesterov(pw[k], grads[step][k].numpy(), buf.get(k), 0.01, 0.937,
                                              wd.get(id(p), 0.0), step == 0)
        d = o_opt.ema_decay_ramp(step + 1)
        for k in ema_o:
            if ema_o[k].dtype.kind == "f":
                cur = pw[k] if k in pw else m.state_dict()[k].numpy()
                ema_o[k] = o_opt.ema_update(ema_o[k], cur, d)
                semi_o[k] = o_opt.ema_update(semi_o[k], ema_o[k], 0.999)
    for k, p in m.named_parameters():
        assert np.allclose(p.detach().numpy(), pw[k], rtol=1e-6, atol=1e-8), k
    esd, ssd = ema.ema.state_dict(), semi.ema.state_dict()
    for k in ema_o:
        if ema_o[k].dtype.kind == "f":
            assert np.allclose(esd[k].numpy(), ema_o[k], rtol=1e-6, atol=1e-8), k
            assert np.allclose(ssd[k].numpy(), semi_o[k], rtol=1e-6, atol=1e-8), k
    keys = ["backbone.stage1.conv.weight", "backbone.stage1.bn.weight", "head.m.0.bias",
            "neck.C1.cv3.conv.weight"]
    out = {}
    for k in keys:
        kk = k.replace(".", "__")
        out["p0__" + kk] = p0[k].numpy()
        out["g0__" + kk] = grads[0][k].numpy()
        out["g1__" + kk] = grads[1][k].numpy()
        out["p2__" + kk] = dict(m.named_parameters())[k].detach().numpy()
        out["ema2__" + kk] = esd[k].numpy()
        out["semi2__" + kk] = ssd[k].numpy()
    out["groups"] = np.array([len(g0), len(g1), len(g2)])
    save("optim", **out)


def case_ssod_step(cfg, model):
    """One real ``SSODTrainer.train_instance`` (trainer/ssod_trainer.py:587-680) + ``update_optimizer``
    (:458-488) of the reference, CPU fp32, on the tiny model.  The trainer object is created without
    running its data/logging set-up (``object.__new__``) and given exactly the attributes the two
    methods read; RANK=1 / WORLD_SIZE=1 skips the rank-0 logging block, not the arithmetic."""
    import copy as _copy
    from torch.cuda import amp
    from models.loss.loss import ComputeLoss, DomainLoss, TargetLoss
    from models.loss.ssod.ssod_loss import ComputeStudentMatchLoss
    from trainer.ssod_trainer import SSODTrainer
    from utils.self_supervised_utils import FairPseudoLabel
    from utils.torch_utils import CosineEMA, ModelEMA
    # The reference's ``tobj[b,a,gj,gi] = v`` scatter (ssod_loss.py:231,248) has duplicate indices; torch's
    # CPU index_put_ is only sequential (= last writer wins, the rule this repo implements) when it does
    # not split the rows across threads, so the reference step is run single-threaded here.
    torch.set_num_threads(1)
    rng = np.random.default_rng(77)
    m = _copy.deepcopy(model).train()
    with torch.no_grad():                      # make the (random) teacher emit detections above 0.1
        for mi in m.head.m:
            b = mi.bias.view(m.head.na, -1)
            b[:, 4] += 6.0
            b[:, 5:] += 3.5
            b[:, 5 + 7] += 2.0
    B, S = 2, 64
    imgs = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    u_ori = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    u_str = np.clip(u_ori + rng.normal(0, 0.05, u_ori.shape), 0, 1).astype(np.float32)
    targets = synth_targets(rng, B, n_per=(2, 5), with_edge=False)
    M_s = np.zeros((B, 13), np.float64)
    M_s[0] = [0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1.0, 0, 0]
    M_s[1] = [1, 0.9, 0, 3.0, 0, 0.9, 2.0, 0, 0, 1, 0.9, 0, 1]
    t = object.__new__(SSODTrainer)
    t.cfg = cfg; t.model = m; t.model_type = 'yolov5'; t.cuda = False; t.device = torch.device('cpu')
    t.RANK = 1; t.WORLD_SIZE = 1; t.extra_teacher_models = []
    t.epochs = cfg.epochs; t.epoch = 0; t.batch_size = cfg.Dataset.batch_size
    t.ema = ModelEMA(m)
    t.semi_ema = CosineEMA(t.ema.ema, decay_start=cfg.SSOD.ema_rate, total_epoch=t.epochs)
    t.pseudo_label_creator = FairPseudoLabel(cfg)
    t.compute_loss = ComputeLoss(m, cfg)
    t.compute_un_sup_loss = ComputeStudentMatchLoss(m, cfg)
    t.domain_loss = DomainLoss(); t.target_loss = TargetLoss()
    t.da_loss_weights = cfg.SSOD.da_loss_weights
    t.fixed_accumulate = cfg.SSOD.fixed_accumulate
    t.scaler = amp.GradScaler(enabled=False)
    # optimizer exactly as trainer/trainer.py:193-243
    t.accumulate = max(round(64 / t.batch_size), 1)
    wd = cfg.hyp.weight_decay * t.batch_size * t.accumulate / 64
    g_bnw, g_w, g_b = [], [], []
    for v in m.modules():
        if hasattr(v, 'bias') and isinstance(v.bias, torch.nn.Parameter):
            g_b.append(v.bias)
        if isinstance(v, torch.nn.BatchNorm2d):
            g_bnw.append(v.weight)
        elif hasattr(v, 'weight') and isinstance(v.weight, torch.nn.Parameter):
            g_w.append(v.weight)
    t.optimizer = torch.optim.SGD(g_b, lr=cfg.hyp.lr0, momentum=cfg.hyp.momentum, nesterov=True)
    t.optimizer.add_param_group({'params': g_w, 'weight_decay': wd})
    t.optimizer.add_param_group({'params': g_bnw})
    t.lf = lambda x: (1 - x / (t.epochs - 1)) * (1.0 - cfg.hyp.lrf) + cfg.hyp.lrf
    t.scheduler = torch.optim.lr_scheduler.LambdaLR(t.optimizer, lr_lambda=t.lf)
    nb = 1000
    t.nw = min(max(round(cfg.hyp.warmup_epochs * nb), 1000), (t.epochs - 0) / 2 * nb)
    t.warmup_bias_lr = cfg.hyp.warmup_bias_lr; t.warmup_momentum = cfg.hyp.warmup_momentum
    t.momentum = cfg.hyp.momentum
    t.last_opt_step = -1
    ni = 500
    t.optimizer.zero_grad()
    captured = {}
    orig = t.compute_un_sup_loss.__call__

    class Spy:
        def __init__(self, inner):
            self.inner = inner
            self.ignore_thres_low, self.ignore_thres_high = inner.ignore_thres_low, inner.ignore_thres_high

        def __call__(self, p, tg):
            captured['targets9'] = tg.detach().clone().numpy()
            loss, items = self.inner(p, tg)
            captured['un_items'] = np.array([float(items[k]) for k in ('ss_box', 'ss_obj', 'ss_cls')], np.float32)
            captured['un_loss'] = loss.detach().numpy().copy()
            return loss, items

    t.compute_un_sup_loss = Spy(t.compute_un_sup_loss)
    closs = t.compute_loss

    def spy_sup(p, tg):
        loss, items = closs(p, tg)
        captured['sup_items'] = np.array([items[k].item() for k in ('box', 'obj', 'cls')], np.float32)
        captured['sup_loss'] = loss.detach().numpy().copy()
        return loss, items

    t.compute_loss = spy_sup
    t.train_instance(torch.from_numpy(imgs), torch.from_numpy(targets), None, torch.from_numpy(u_str),
                     torch.from_numpy(u_ori), None, torch.from_numpy(M_s), ni, None, None)
    assert 'targets9' in captured and captured['targets9'].shape[0] > 0, "no pseudo labels: bump the biases"
    print("   pseudo labels:", captured['targets9'].shape[0], "sup", captured['sup_loss'], "unsup", captured['un_loss'])
    out = dict(imgs=imgs, u_ori=u_ori, u_str=u_str, targets=targets, M_s=M_s, ni=np.int64(ni), nb=np.int64(nb),
               **captured)
    keys = ["backbone.stage1.conv.weight", "backbone.stage1.bn.weight", "backbone.stage3_2.m.0.cv2.conv.weight",
            "neck.C2.cv3.bn.bias", "head.m.1.weight", "head.m.2.bias", "backbone.sppf.cv2.bn.running_var",
            "det_8.conv1.weight"]
    sd, esd, ssd = m.state_dict(), t.ema.ema.state_dict(), t.semi_ema.ema.state_dict()
    sd0 = model.state_dict()
    for k in keys:
        kk = k.replace(".", "__")
        out["m__" + kk] = sd[k].numpy(); out["e__" + kk] = esd[k].numpy(); out["s__" + kk] = ssd[k].numpy()
        out["d__" + kk] = (sd[k] - sd0[k]).numpy() if k in sd0 and "head" not in k else np.zeros(1)
    out["lrs"] = np.array([g['lr'] for g in t.optimizer.param_groups])
    out["moms"] = np.array([g['momentum'] for g in t.optimizer.param_groups])
    save("ssod_step", **out)


def case_v8():
    """YOLOv8 anchor-free path from the LIVE reference: tiny-width model (C2f backbone + neck + decoupled
    head) train / eval outputs, and TaskAlignedAssigner on synthetic predictions -- one case small, one at 8400 anchors
    with crowded gts (anchors claimed by several gts, padded gt rows, metric-0 candidates)."""
    from models.assigner.tal_assigner import TaskAlignedAssigner
    from models.detector.yolo import Model
    from . import v8
    cfg = ref_loader.get_cfg("configs/sup/public/yolov8m_coco.yaml", ["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33])
    cfg.freeze()
    torch.manual_seed(0)
    m = Model(cfg)
    with torch.no_grad():                      # non-trivial BN statistics / affine so that eval mode is a real test
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.1)
    o = v8.Model.from_cfg(cfg)
    o.load_state_dict(m.state_dict(), strict=True)
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.uniform(0, 1, (2, 3, 64, 64)).astype(np.float32))
    out = {"w__" + k.replace(".", "__"): v.numpy() for k, v in m.state_dict().items()}
    m.train(); o.train()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    f, c, r = m(x)
    m.load_state_dict(sd0); o.load_state_dict(sd0)            # the train forward moved the running statistics
    f2, c2, r2 = o(x)
    assert torch.equal(c, c2) and torch.equal(r, r2), "oracle v8 train forward differs from the reference"
    o.load_state_dict(sd0)
    m.eval(); o.eval()
    z, _ = m(x)
    z2, _ = o(x)
    assert (z - z2).abs().max().item() < 2e-4
    out.update(x=x.numpy(), train_cls=c.detach().numpy(), train_reg=r.detach().numpy(), eval_z=z.detach().numpy(),
               feat_shapes=np.array([t.shape[-2:] for t in f]))
    save("v8_model", **out)
    # ---- assigner ----
    asg = TaskAlignedAssigner(13, 80, 1.0, 6.0)
    tal = {}
    for name, (B, shapes, G, crowd) in dict(small=(2, [(8, 8), (4, 4), (2, 2)], 5, False), full=(2, [(80, 80), (40, 40), (20, 20)], 24, True)).items():
        pts, _ = v8.anchor_points_train(shapes, (8, 16, 32))
        A = pts.shape[0]
        S = shapes[0][0] * 8
        g = torch.Generator().manual_seed(5 + B)
        ps = torch.rand(B, A, 80, generator=g) ** 3
        ctr = pts.unsqueeze(0).expand(B, A, 2) + torch.randn(B, A, 2, generator=g) * 6
        wh = torch.rand(B, A, 2, generator=g) * (S / 3) + 4
        pb = torch.cat([ctr - wh / 2, ctr + wh / 2], -1)
        gl = torch.randint(0, 80, (B, G, 1), generator=g).float()
        lo, hi = (0.35, 0.65) if crowd else (0.2, 0.8)       # crowded: gts pile up in the middle -> multi-claims
        gc = (torch.rand(B, G, 2, generator=g) * (hi - lo) + lo) * S
        gwh = torch.rand(B, G, 2, generator=g) * (S / 3) + S / 16
        gb = torch.cat([gc - gwh / 2, gc + gwh / 2], -1).clamp(0, S)
        gb[-1, G - 2:] = 0; gl[-1, G - 2:] = -1              # padded rows as ComputeTalLoss.preprocess makes them
        mg = (gb.sum(-1, keepdim=True) > 0).float()
        tl, tb, ts, fg = asg(ps, pb, pts, gl, gb, mg)
        mine = v8.tal_assign(ps, pb, pts, gl, gb, mg)
        eff = ts.sum(-1) > 0
        assert torch.equal(ts, mine[2]) and torch.equal(fg, mine[3]) and torch.equal(tl[eff], mine[0][eff]), name
        print(f"   tal[{name}]: A={A} fg={int(fg.sum())} effective={int(eff.sum())} multi-claimed handled")
        for k, v in dict(ps=ps, pb=pb, pts=pts, gl=gl, gb=gb, mg=mg, tl=tl, tb=tb, ts=ts, fg=fg).items():
            tal[f"{name}__{k}"] = v.numpy()
    save("v8_tal", **tal)


def ota_inputs(seed=7, B=2, nc=6, shapes=((80, 80), (40, 40), (20, 20))):
    """seeded inputs of the SimOTA case: regenerated (not stored) by tests/test_ota.py"""
    rng = np.random.default_rng(seed)
    t = synth_targets(rng, B, n_per=(3, 12))
    t[:, 1] = rng.integers(0, nc, t.shape[0])
    p = [rng.normal(0, 1.5, (B, 3, ny, nx, 5 + nc)).astype(np.float32) for ny, nx in shapes]
    # make the predicted boxes of some cells plausible (random logits give IoUs near 0 and dynamic_k == 1 everywhere)
    for pi in p:
        pi[..., :4] *= 0.3
    return t, p


def case_ota():
    """ComputeLoss with Loss.assigner_type == 'SimOTA' (models/loss/loss.py:210-303, yolo_anchor_assigner.py:104-317)."""
    nc = 6
    cfg = ref_loader.get_cfg(SSOD_YAML, TINY + ["Dataset.nc", nc, "Loss.assigner_type", "SimOTA"])
    cfg.freeze()
    from models.detector.yolo_ssod import Model
    from models.loss.loss import ComputeLoss
    torch.manual_seed(0)
    model = Model(cfg)
    closs = ComputeLoss(model, cfg)
    assert closs.ota and closs.top_k == 13
    anchors = model.head.anchors.clone()
    strides = [float(s) for s in model.head.stride]
    t, p = ota_inputs(nc=nc)
    tt = torch.from_numpy(t)
    pr = [torch.from_numpy(x).requires_grad_(True) for x in p]
    loss, items = closs(pr, tt)
    loss.backward()
    bs, as_, gjs, gis, ota_t, anch = closs.ota_assigner([x.detach() for x in pr], tt)
    kw = dict(nc=nc, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w, anchor_t=closs.anchor_t)
    mine = o_loss.build_ota_targets([x.detach() for x in pr], tt, anchors, strides, nc=nc, anchor_t=closs.anchor_t)
    out = dict(targets=t, anchors=anchors.numpy(), strides=np.array(strides), nc=np.int64(nc))
    npos = 0
    for i in range(3):
        m = mine[i]
        assert torch.equal(bs[i], m["b"]) and torch.equal(as_[i], m["a"]) and torch.equal(gjs[i], m["gj"]) \
            and torch.equal(gis[i], m["gi"]) and torch.equal(ota_t[i], m["target"]) and torch.equal(anch[i], m["anch"]), i
        npos += m["b"].shape[0]
        for k in ("b", "a", "gj", "gi", "slot"):
            out[f"l{i}_{k}"] = m[k].numpy()
        out[f"l{i}_target"] = m["target"].numpy()
    dyn_gt1 = sum(int((np.unique(mine[i]["target"].numpy(), axis=0, return_counts=True)[1] > 1).sum()) for i in range(3))
    print(f"  SimOTA positives {npos}; target rows used more than once on a level: {dyn_gt1}")
    po = [torch.from_numpy(x).requires_grad_(True) for x in p]
    loss2, items2 = o_loss.ota_loss(po, tt, anchors, strides, **kw)
    loss2.backward()
    assert torch.allclose(loss, loss2, rtol=1e-6, atol=1e-7), (loss, loss2)
    for a, b in zip(pr, po):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8)
    out["loss"] = loss.detach().numpy()
    out["items"] = np.array([items[k].item() for k in ("box", "obj", "cls")], np.float32)
    out["weights"] = np.array([closs.box_w, closs.obj_w, closs.cls_w, closs.anchor_t], np.float64)
    out["grad2"] = pr[2].grad.numpy()
    for i in range(2):          # levels 0/1: the gradient rows of the SimOTA positives and per-channel sums
        m = mine[i]
        out[f"gradrows{i}"] = pr[i].grad[m["b"], m["a"], m["gj"], m["gi"]].numpy()
        out[f"gradsum{i}"] = pr[i].grad.double().sum((0, 1, 2, 3)).numpy()
    save("ota", **out)


def labelmatch_pred(rng, B, A, nc):
    pred = synth_pred(rng, B, A, nc, obj_pow=2, cls_pow=6)
    pred[..., 5:] = 0
    hot = rng.integers(0, nc, (B, A))
    np.put_along_axis(pred[..., 5:], hot[..., None], rng.uniform(0.3, 1, (B, A, 1)).astype(np.float32), 2)
    return pred


def case_labelmatch():
    """utils/labelmatch.py: LabelMatch.create_pseudo_label_online_with_gt over two epochs of batches + update_epoch_cls_thr"""
    import io
    from contextlib import redirect_stdout
    from utils.labelmatch import LabelMatch
    from . import labelmatch as o_lm
    nc = 5
    cfg = ref_loader.get_cfg(SSOD_YAML, TINY + ["Dataset.nc", nc, "SSOD.pseudo_label_type", "LabelMatch",
                                                "SSOD.resample_low_percent", 0.3, "SSOD.resample_high_percent", 0.1,
                                                "Dataset.names", [str(i) for i in range(nc)]])
    cfg.freeze()
    rng = np.random.default_rng(33)
    B, A, W, H = 3, 400, 640, 640
    M_s = np.zeros((B, 13), np.float64)
    for i in range(B):
        s = [1.0, 0.7, 1.3][i]
        M = np.array([[s, 0.03 * i, 40.0 * i - 30], [-0.02 * i, s, 25.0 * i - 10], [0, 0, 1]], np.float64)
        M_s[i] = [i, *M.reshape(-1), s, i % 2, (i + 1) % 2]
    imgs = torch.zeros(B, 3, H, W)
    lm = LabelMatch(cfg, 100, 5, cls_ratio_gt=np.full(nc, 1.0 / nc))
    mine = o_lm.LabelMatchState(nc, cfg.SSOD.ignore_thres_low, cfg.SSOD.ignore_thres_high, cfg.SSOD.resample_high_percent,
                                cfg.SSOD.resample_low_percent)
    out = dict(M_s=M_s, hw=np.array([H, W]), nc=np.int64(nc), seed=np.int64(33), BA=np.array([B, A]),
               thr=np.array([cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres, cfg.SSOD.ignore_thres_low, cfg.SSOD.ignore_thres_high,
                             cfg.SSOD.resample_low_percent, cfg.SSOD.resample_high_percent]))
    step = 0
    for epoch, nb in enumerate((3, 2)):
        for _ in range(nb):
            pred = labelmatch_pred(rng, B, A, nc)
            with redirect_stdout(io.StringIO()):
                ref_t, ref_inv = lm.create_pseudo_label_online_with_gt(torch.from_numpy(pred.copy()), imgs, torch.from_numpy(M_s),
                                                                       imgs.clone())
            mt, minv = mine.create_pseudo_label(pred, M_s, W, H, cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
            assert ref_inv == minv and np.array_equal(ref_t.numpy(), mt), "LabelMatch pseudo-label pin failed"
            out[f"targets{step}"] = mt
            step += 1
        for c in range(nc):
            assert sorted(lm.score_list_epoch[c]) == sorted(mine.score_list_epoch[c])
        with redirect_stdout(io.StringIO()):
            lm.update_epoch_cls_thr(epoch)
        mine.update_epoch_cls_thr(epoch)
        assert list(lm.cls_thr_high) == list(mine.cls_thr_high) and list(lm.cls_thr_low) == list(mine.cls_thr_low), \
            (lm.cls_thr_high, mine.cls_thr_high, lm.cls_thr_low, mine.cls_thr_low)
        out[f"thr_high{epoch}"] = np.array(lm.cls_thr_high, np.float64)
        out[f"thr_low{epoch}"] = np.array(lm.cls_thr_low, np.float64)
        print(f"  epoch {epoch}: high {np.round(lm.cls_thr_high, 4)} low {np.round(lm.cls_thr_low, 4)}")
    save("labelmatch", **out)


def case_focal():
    """ComputeLoss with Loss.fl_gamma = 1.5 (FocalLoss around BCEcls / BCEobj, models/loss/loss.py:37-62, :112-114) on the inputs
    of tests/golden/compute_loss.npz"""
    g = np.load(os.path.join(OUT, "compute_loss.npz"))
    cfg = ref_loader.get_cfg(SSOD_YAML, TINY + ["Loss.fl_gamma", 1.5, "Loss.label_smoothing", 0.1])
    cfg.freeze()
    from models.detector.yolo_ssod import Model
    from models.loss.loss import ComputeLoss
    torch.manual_seed(0)
    model = Model(cfg)
    closs = ComputeLoss(model, cfg)
    anchors = torch.from_numpy(g["anchors"])
    assert torch.equal(model.head.anchors, anchors)
    t = torch.from_numpy(g["targets"])
    pr = [torch.from_numpy(g[f"p{i}"]).requires_grad_(True) for i in range(3)]
    loss, items = closs(pr, t)
    loss.backward()
    po = [torch.from_numpy(g[f"p{i}"]).requires_grad_(True) for i in range(3)]
    loss2, _ = o_loss.compute_loss(po, t, anchors, nc=80, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w,
                                   anchor_t=closs.anchor_t, cp=closs.cp, cn=closs.cn, fl_gamma=1.5)
    loss2.backward()
    assert torch.allclose(loss, loss2, rtol=1e-6, atol=1e-7), (loss, loss2)
    for a, b in zip(pr, po):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-9)
    out = dict(loss=loss.detach().numpy(), items=np.array([items[k].item() for k in ("box", "obj", "cls")], np.float32),
               hp=np.array([1.5, 0.1, closs.cp, closs.cn], np.float64))
    for i in range(3):
        out[f"grad{i}"] = pr[i].grad.numpy()
    save("focal_loss", **out)


def case_autobalance():
    """ComputeLoss with Loss.autobalance (models/loss/loss.py:118, :193-197): three consecutive calls, the balance weights evolve"""
    g = np.load(os.path.join(OUT, "compute_loss.npz"))
    cfg = ref_loader.get_cfg(SSOD_YAML, TINY + ["Loss.autobalance", True])
    cfg.freeze()
    from models.detector.yolo_ssod import Model
    from models.loss.loss import ComputeLoss
    torch.manual_seed(0)
    model = Model(cfg)
    closs = ComputeLoss(model, cfg)
    assert closs.autobalance and closs.ssi == 1
    anchors = torch.from_numpy(g["anchors"])
    t = torch.from_numpy(g["targets"])
    bal = [4.0, 1.0, 0.4]
    out = {}
    for k in range(3):
        sc = 1.0 + 0.25 * k
        pr = [(torch.from_numpy(g[f"p{i}"]) * sc).requires_grad_(True) for i in range(3)]
        loss, items = closs(pr, t)
        loss.backward()
        po = [(torch.from_numpy(g[f"p{i}"]) * sc).requires_grad_(True) for i in range(3)]
        loss2, _ = o_loss.compute_loss(po, t, anchors, nc=80, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w,
                                       anchor_t=closs.anchor_t, balance=bal, autobalance_ssi=1)
        loss2.backward()
        assert torch.allclose(loss, loss2, rtol=1e-6, atol=1e-7), (k, loss, loss2)
        assert np.allclose(closs.balance, bal, rtol=1e-6), (closs.balance, bal)
        out[f"loss{k}"] = loss.detach().numpy()
        out[f"balance{k}"] = np.array(closs.balance, np.float64)
        out[f"grad{k}"] = pr[0].grad.numpy()[..., 4].copy()             # objectness-channel gradient of level 0 carries balance[0]
    save("autobalance", **out)


def case_mosaic():
    """load_mosaic_with_M (utils/datasets_ssod.py:732-792) run by the LIVE reference on a small in-memory dataset: what it pastes
    where (the 2s x 2s canvas it hands to cv2.resize), the labels it hands to random_perspective_with_M, and how much of the
    `random` stream it consumes.  cv2 is absent: `cv2.resize` is replaced by a recorder that returns the 2:1 box average (the
    restated INTER_AREA fast path OpenCV takes for this call -- the PIXELS of that step are therefore this package's restatement,
    unpinned), `random_perspective_with_M` by a recorder (the strong view is a separate stage: StrongViewGenerator)."""
    import random as pyrandom
    import types
    import utils.datasets_ssod as D
    s = 32
    rng = np.random.default_rng(5)
    shapes = [(32, 24), (20, 32), (32, 32), (17, 32), (32, 13), (28, 32)]          # load_image: longest side = img_size
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    labels = []
    for k in range(len(shapes)):
        n = int(rng.integers(0, 4))
        lb = np.zeros((n, 5), np.float32)
        lb[:, 0] = rng.integers(0, 80, n)
        lb[:, 1:3] = rng.uniform(0.2, 0.8, (n, 2))
        lb[:, 3:5] = rng.uniform(0.05, 0.5, (n, 2))
        labels.append(lb)
    ds = types.SimpleNamespace(img_size=s, mosaic_border=[-s // 2, -s // 2], indices=list(range(len(shapes))), imgs=imgs,
                               img_hw0=shapes, img_hw=shapes, labels=labels, segments=[[] for _ in shapes],
                               hyp=dict(degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0))
    rec = {}

    def resize(img, dsize, **k):
        rec.setdefault("canvas", []).append(img.copy())
        assert img.shape[0] == 2 * dsize[1] and img.shape[1] == 2 * dsize[0]
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def perspective(img, targets=(), segments=(), **k):
        rec.setdefault("weak", []).append(img.copy())
        rec.setdefault("labels4", []).append(np.array(targets, dtype=np.float64).reshape(-1, 5).copy())
        return img, targets, np.zeros(13)
    keep = (D.cv2.__dict__.get("resize"), D.random_perspective_with_M)
    D.cv2.resize = resize
    D.random_perspective_with_M = perspective
    out = {}
    try:
        pyrandom.seed(1234)
        for j, index in enumerate((0, 3, 5, 2)):
            D.load_mosaic_with_M(ds, index)
        out["next_random"] = np.array([pyrandom.random()])          # the stream position after four mosaics
    finally:
        if keep[0] is None:
            del D.cv2.resize
        else:
            D.cv2.resize = keep[0]
        D.random_perspective_with_M = keep[1]
    for k, im in enumerate(imgs):
        out[f"img{k}"] = im
        out[f"lab{k}"] = labels[k]
    for j in range(4):
        out[f"canvas{j}"] = rec["canvas"][j]
        out[f"weak{j}"] = rec["weak"][j]
        out[f"labels4_{j}"] = rec["labels4"][j]
    out["index"] = np.array([0, 3, 5, 2])
    out["s"] = np.array([s])
    save("mosaic", **out)


def main():
    if not ref_loader.available():
        sys.exit("reference tree not present; golden vectors can only be generated in the build container")
    ref_loader.load()
    if len(sys.argv) > 1 and sys.argv[1] == "v8":
        print("== YOLOv8 path")
        case_v8()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "autobalance":
        print("== autobalance")
        case_autobalance()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "focal":
        print("== focal loss")
        case_focal()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "labelmatch":
        print("== LabelMatch")
        case_labelmatch()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mosaic":
        print("== mosaic")
        case_mosaic()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "nms":
        print("== nms")
        case_nms()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "nms_options":
        print("== non_max_suppression_ssod options")
        case_nms_ssod_options()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ota":
        print("== SimOTA loss")
        case_ota()
        return
    torch.set_num_threads(4)
    print("nms ..."); case_nms(); case_nms_ssod_options()
    print("assigner / losses ..."); cfg, model = case_assigner_and_losses()
    print("pseudo label ..."); case_pseudo_label(cfg)
    print("model ..."); case_model(cfg, model)
    print("optimizer / EMA ..."); case_optim(model)
    print("ssod step (reference SSODTrainer.train_instance) ..."); case_ssod_step(cfg, model)
    print("SimOTA loss ..."); case_ota()
    print("LabelMatch ..."); case_labelmatch()
    print("focal loss ..."); case_focal()
    print("autobalance ..."); case_autobalance()
    print("mosaic ..."); case_mosaic()
    print("done")


if __name__ == "__main__":
    main()
