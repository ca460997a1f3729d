"""TEST INFRASTRUCTURE -- plain-torch fp32 restatement of the detection losses.

Restates (CPU, fp32, autograd gives the reference gradients):
  * ``bbox_iou(..., x1y1x2y2=False, CIoU=True)``   utils/metrics.py:207-245
  * ``ComputeLoss.default_loss``                   models/loss/loss.py:138-208
  * ``ComputeStudentMatchLoss.select_targets`` / ``default_loss``
                                                   models/loss/ssod/ssod_loss.py:130-296
  * ``DomainLoss`` / ``TargetLoss`` / ``DomainFocalLoss``
                                                   models/loss/loss.py:312-418

Duplicate-cell rule for ``tobj[b,a,gj,gi] = v`` (loss.py:172, ssod_loss.py:231,248):
the reference's ``index_put_`` is last-writer-wins on CPU; this restatement makes it
explicit (later row in assigner order wins) -- SURVEY.md appendix C.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import assigner as _asg


def ciou_xywh(pbox, tbox, eps=1e-7):
    """metrics.py:207-245 with box1=pbox.T (xywh), box2=tbox (xywh); returns (n,)."""
    b1_x1, b1_x2 = pbox[:, 0] - pbox[:, 2] / 2, pbox[:, 0] + pbox[:, 2] / 2
    b1_y1, b1_y2 = pbox[:, 1] - pbox[:, 3] / 2, pbox[:, 1] + pbox[:, 3] / 2
    b2_x1, b2_x2 = tbox[:, 0] - tbox[:, 2] / 2, tbox[:, 0] + tbox[:, 2] / 2
    b2_y1, b2_y2 = tbox[:, 1] - tbox[:, 3] / 2, tbox[:, 1] + tbox[:, 3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def _assign(p, anchors, targets, anchor_t, with_score=False):
    shapes = [(pi.shape[2], pi.shape[3]) for pi in p]
    res = _asg.build_targets(shapes, anchors.detach().cpu().numpy(),
                             targets.detach().cpu().numpy(), anchor_t, with_score)
    dev = p[0].device
    out = []
    for r in res:
        out.append({k: torch.from_numpy(v).to(dev) for k, v in r.items()})
    return out


def _scatter_last_wins(tobj, r, vals):
    """tobj[b,a,gj,gi] = vals with an explicit later-row-wins rule."""
    flat = ((r["b"] * tobj.shape[1] + r["a"]) * tobj.shape[2] + r["gj"]) * tobj.shape[3] + r["gi"]
    tf = tobj.view(-1)
    for k in range(flat.shape[0]):
        tf[flat[k]] = vals[k]


def _box_cls_terms(pi, r, nc, cp, cn, want_box=True, want_cls=True, fl_gamma=0.0):
    ps = pi[r["b"], r["a"], r["gj"], r["gi"]]
    lbox = lcls = None
    iou = None
    if want_box:
        pxy = ps[:, :2].sigmoid() * 2. - 0.5
        pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * r["anch"]
        iou = ciou_xywh(torch.cat((pxy, pwh), 1), r["tbox"])
        lbox = (1.0 - iou).mean()
    if want_cls and nc > 1:
        t = torch.full_like(ps[:, 5:], cn)
        t[torch.arange(ps.shape[0]), r["tcls"]] = cp
        lcls = focal_bce(ps[:, 5:], t, fl_gamma) if fl_gamma > 0 else F.binary_cross_entropy_with_logits(ps[:, 5:], t)
    return lbox, lcls, iou


def focal_bce(pred, true, gamma, alpha=0.25):
    """FocalLoss(nn.BCEWithLogitsLoss(), gamma) with mean reduction (models/loss/loss.py:37-62)"""
    loss = F.binary_cross_entropy_with_logits(pred, true, reduction="none")
    prob = torch.sigmoid(pred)
    p_t = true * prob + (1 - true) * (1 - prob)
    loss = loss * (true * alpha + (1 - true) * (1 - alpha)) * (1.0 - p_t) ** gamma
    return loss.mean()


def compute_loss(p, targets, anchors, *, nc=80, box_w=0.05, obj_w=1.0, cls_w=0.5,
                 anchor_t=4.0, balance=(4.0, 1.0, 0.4), gr=1.0, cp=1.0, cn=0.0, fl_gamma=0.0, autobalance_ssi=None):
    """... autobalance_ssi: Loss.autobalance (loss.py:193-197) -- `balance` must then be a LIST, updated in place"""
    return _compute_loss(p, targets, anchors, nc, box_w, obj_w, cls_w, anchor_t, balance, gr, cp, cn, fl_gamma, autobalance_ssi)


def _compute_loss(p, targets, anchors, nc, box_w, obj_w, cls_w, anchor_t, balance, gr, cp, cn, fl_gamma, autobalance_ssi):
    """loss.py:138-208.  p: list of (B,na,ny,nx,5+nc); targets (n,6).
    Returns (loss*bs [1], dict(box,obj,cls,loss))."""
    dev = p[0].device
    lcls, lbox, lobj = (torch.zeros(1, device=dev) for _ in range(3))
    asg = _assign(p, anchors, targets, anchor_t)
    for i, pi in enumerate(p):
        r = asg[i]
        tobj = torch.zeros_like(pi[..., 0])
        if r["b"].shape[0]:
            lb, lc, iou = _box_cls_terms(pi, r, nc, cp, cn, fl_gamma=fl_gamma)
            lbox = lbox + lb
            _scatter_last_wins(tobj, r, (1.0 - gr) + gr * iou.detach().clamp(0))
            if lc is not None:
                lcls = lcls + lc
        obji = (focal_bce(pi[..., 4], tobj, fl_gamma) if fl_gamma > 0 else
                F.binary_cross_entropy_with_logits(pi[..., 4], tobj))
        lobj = lobj + obji * balance[i]
        if autobalance_ssi is not None:
            balance[i] = balance[i] * 0.9999 + 0.0001 / obji.detach().item()
    if autobalance_ssi is not None:
        ref = balance[autobalance_ssi]
        balance[:] = [x / ref for x in balance]
    lbox, lobj, lcls = lbox * box_w, lobj * obj_w, lcls * cls_w
    bs = p[0].shape[0]
    loss = lbox + lobj + lcls
    return loss * bs, dict(box=lbox, obj=lobj, cls=lcls, loss=loss * bs)


def select_targets(targets, thr_low, thr_high, with_obj=True):
    """ssod_loss.py:130-192.  targets (N,9) [img,cls,x,y,w,h,conf,obj,clsconf] (any float
    dtype; the reference compares the fp64 values then casts rows to fp32).
    Returns reliable(n,7), uncertain(n,7), uc_obj(n,7), uc_cls(n,7) fp32 arrays."""
    t = np.asarray(targets, np.float64).reshape(-1, 9)
    rel, unc, uobj, ucls = [], [], [], []
    for row in t:
        c = int(row[1])
        if row[6] >= thr_high[c]:
            rel.append(row[:7])
        elif row[6] >= thr_low[c]:
            if with_obj:
                u = np.concatenate((row[:6], row[7:8]))
                unc.append(u)
                if row[7] >= 0.99:
                    uobj.append(u)
                if row[8] >= 0.99:
                    ucls.append(u)
            else:
                unc.append(row[:7])
    f = lambda l: np.asarray(l, np.float64).astype(np.float32).reshape(-1, 7)
    return f(rel), f(unc), f(uobj), f(ucls)


def compute_student_match_loss(p, targets9, anchors, *, nc=80, box_w=0.05, obj_w=0.7, cls_w=0.3,
                               anchor_t=4.0, balance=(4.0, 1.0, 0.4), gr=1.0, cp=1.0, cn=0.0,
                               thr_low=None, thr_high=None, ignore_obj=False, with_obj=True,
                               with_bbox=True, with_cls=False):
    """ssod_loss.py:194-288 (uncertain_aug path == the other path, :198-207)."""
    dev = p[0].device
    thr_low = thr_low if thr_low is not None else [0.1] * nc
    thr_high = thr_high if thr_high is not None else [0.6] * nc
    lcls, lbox, lobj = (torch.zeros(1, device=dev) for _ in range(3))
    rel, unc, uobj, ucls = select_targets(targets9.detach().cpu().numpy(), thr_low, thr_high, with_obj)
    tt = lambda a: torch.from_numpy(a).to(dev)
    asg = _assign(p, anchors, tt(rel), anchor_t)
    uc = _assign(p, anchors, tt(unc), anchor_t, with_score=True)
    uco = _assign(p, anchors, tt(uobj), anchor_t, with_score=True)
    ucc = _assign(p, anchors, tt(ucls), anchor_t, with_score=True)
    for i, pi in enumerate(p):
        r = asg[i]
        tobj = torch.zeros_like(pi[..., 0])
        if r["b"].shape[0]:
            lb, lc, iou = _box_cls_terms(pi, r, nc, cp, cn)
            lbox = lbox + lb
            _scatter_last_wins(tobj, r, (1.0 - gr) + gr * iou.detach().clamp(0))
            if lc is not None:
                lcls = lcls + lc
        u = uc[i]
        if u["b"].shape[0]:                                           # :243-248
            vals = torch.full_like(u["tscore"], -1.0) if ignore_obj else u["tscore"]
            _scatter_last_wins(tobj, u, vals)
        if with_bbox and uco[i]["b"].shape[0]:                        # :250-263
            lb, _, _ = _box_cls_terms(pi, uco[i], nc, cp, cn, want_cls=False)
            lbox = lbox + lb
        if with_cls and ucc[i]["b"].shape[0]:                         # :266-275
            _, lc, _ = _box_cls_terms(pi, ucc[i], nc, cp, cn, want_box=False)
            if lc is not None:
                lcls = lcls + lc
        valid = tobj >= 0                                             # :277-278
        lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4][valid], tobj[valid]) * balance[i]
    lbox, lobj, lcls = lbox * box_w, lobj * obj_w, lcls * cls_w
    bs = p[0].shape[0]
    loss = lbox + lobj + lcls
    return loss * bs, dict(ss_box=lbox, ss_obj=lobj, ss_cls=lcls)


def domain_loss(features, label):
    """DomainLoss (label 0, loss.py:398-421) / TargetLoss (label 1, loss.py:376-395):
    features = 3 x (B,2,H,W) netD logits -> permute(0,2,3,1).reshape(-1,2), concatenated;
    0.5 * mean(-(1-p_label)^2 * log p_label), p = softmax over the 2 logits
    (DomainFocalLoss, loss.py:312-368, alpha=1, gamma=2, size_average)."""
    x = torch.cat([f.permute(0, 2, 3, 1).reshape(-1, 2) for f in features], 0)
    p = torch.softmax(x, dim=1)[:, label]
    return 0.5 * (-(1 - p) ** 2 * p.log()).mean()


# ---- SimOTA (Loss.assigner_type == 'SimOTA') --------------------------------------------------------
def box_iou_xyxy(b1, b2):
    """utils/metrics.py:252-274: (N,4) x (M,4) -> (N,M), no epsilon."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (a1[:, None] + a2 - inter)


def _xywh2xyxy(x):
    return torch.stack((x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2, x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2), 1)


def build_ota_targets(p, targets, anchors, strides, *, nc, anchor_t=4.0, top_k=13, img_size=640.0):
    """YOLOAnchorAssigner.build_ota_targets (models/assigner/yolo_anchor_assigner.py:104-264) over the candidates of
    find_3_positive (:266-317, the same rows as build_targets).  Returns per level a dict with b, a, gj, gi (int64), anch
    (n,2), target (n,6: the matched target row) and slot (n: index of the candidate in the level's find_3_positive list) --
    rows in the reference's order (image-major, then candidate order)."""
    with torch.no_grad():
        cand = _assign(p, anchors, targets, anchor_t)
        nl = len(p)
        out = [dict(b=[], a=[], gj=[], gi=[], anch=[], target=[], slot=[]) for _ in range(nl)]
        for b in range(p[0].shape[0]):
            sel = targets[:, 0] == b                                            # :123
            tt = targets[sel]
            if tt.shape[0] == 0:
                continue
            txyxy = _xywh2xyxy(tt[:, 2:6] * img_size)                           # :128-129 (the literal 640)
            rows = []
            for i, pi in enumerate(p):
                r = cand[i]
                idx = torch.nonzero(r["b"] == b).flatten()
                fg = pi[r["b"][idx], r["a"][idx], r["gj"][idx], r["gi"][idx]].float()
                grid = torch.stack((r["gi"][idx], r["gj"][idx]), 1).float()
                pxy = (fg[:, :2].sigmoid() * 2. - 0.5 + grid) * strides[i]     # :160
                pwh = (fg[:, 2:4].sigmoid() * 2) ** 2 * r["anch"][idx] * strides[i]
                rows.append(dict(level=torch.full((idx.shape[0],), i), idx=idx, box=_xywh2xyxy(torch.cat((pxy, pwh), 1)),
                                 cls=fg[:, 5:5 + nc], e2e=fg[:, -1:]))       # :155-157: p_obj_e2e is the LAST channel
            level = torch.cat([r_["level"] for r_ in rows])
            idx = torch.cat([r_["idx"] for r_ in rows])
            box = torch.cat([r_["box"] for r_ in rows])
            if box.shape[0] == 0:
                continue
            pcls = torch.cat([r_["cls"] for r_ in rows])
            pe2e = torch.cat([r_["e2e"] for r_ in rows])
            iou = box_iou_xyxy(txyxy, box)                                      # :181
            iou_loss = -torch.log(iou + 1e-8)
            topv, _ = torch.topk(iou, min(top_k, iou.shape[1]), dim=1)
            dyn = torch.clamp(topv.sum(1).int(), min=1)                         # :186
            onehot = F.one_hot(tt[:, 1].to(torch.int64), nc).float()            # (G, nc)
            y = (pcls.sigmoid() * pe2e.sigmoid()).sqrt()                        # (M, nc)   :196-202
            logit = torch.log(y / (1 - y))
            cls_loss = F.binary_cross_entropy_with_logits(logit[None].expand(tt.shape[0], -1, -1),
                                                          onehot[:, None, :].expand(-1, y.shape[0], -1),
                                                          reduction="none").sum(-1)
            cost = cls_loss + 3.0 * iou_loss                                    # :208-211
            G, M = cost.shape
            mark = torch.zeros((G, M), dtype=torch.bool)
            order = torch.arange(M)
            for g in range(G):                                                  # :215-219, ties: smaller candidate index
                c = cost[g].clone()
                c[torch.isnan(c)] = float("inf")
                key = sorted(range(M), key=lambda j: (float(c[j]), j))
                mark[g, key[:int(dyn[g])]] = True
            n_per = mark.sum(0)
            multi = n_per > 1
            if multi.any():                                                     # :223-226
                amin = torch.argmin(torch.where(torch.isnan(cost), torch.full_like(cost, float("inf")), cost)[:, multi], 0)
                mark[:, multi] = False
                mark[amin, order[multi]] = True
            fg_mask = mark.any(0)
            gt_of = mark[:, fg_mask].float().argmax(0)                          # :228
            for i in range(nl):
                m = (level[fg_mask] == i)
                ci = idx[fg_mask][m]
                r = cand[i]
                o = out[i]
                o["b"].append(r["b"][ci]); o["a"].append(r["a"][ci]); o["gj"].append(r["gj"][ci]); o["gi"].append(r["gi"][ci])
                o["anch"].append(r["anch"][ci]); o["target"].append(tt[gt_of[m]]); o["slot"].append(ci)
        res = []
        for o in out:
            cat = lambda k, shape, dt: torch.cat(o[k]) if o[k] else torch.zeros(shape, dtype=dt)
            res.append(dict(b=cat("b", (0,), torch.int64), a=cat("a", (0,), torch.int64), gj=cat("gj", (0,), torch.int64),
                            gi=cat("gi", (0,), torch.int64), anch=cat("anch", (0, 2), torch.float32),
                            target=cat("target", (0, targets.shape[1]), torch.float32), slot=cat("slot", (0,), torch.int64)))
        return res


def ota_loss(p, targets, anchors, strides, *, nc=80, box_w=0.05, obj_w=1.0, cls_w=0.5, anchor_t=4.0,
             balance=(4.0, 1.0, 0.4), gr=1.0, cp=1.0, cn=0.0, top_k=13, img_size=640.0):
    """ComputeLoss.ota_loss (models/loss/loss.py:210-303): the SimOTA-matched half (objectness read from the LAST channel,
    :246) plus the plain build_targets half (:251-292), summed before the weights."""
    dev = p[0].device
    lcls, lbox, lobj = (torch.zeros(1, device=dev) for _ in range(3))
    ota = build_ota_targets(p, targets, anchors, strides, nc=nc, anchor_t=anchor_t, top_k=top_k, img_size=img_size)
    for i, pi in enumerate(p):
        r = ota[i]
        tobj = torch.zeros_like(pi[..., 0])
        if r["b"].shape[0]:
            ny, nx = pi.shape[2], pi.shape[3]
            gain = torch.tensor([nx, ny, nx, ny], dtype=torch.float32)
            tbox = r["target"][:, 2:6] * gain                                   # :231
            tbox[:, :2] -= torch.stack((r["gi"], r["gj"]), 1).float()           # :232
            rr = dict(r, tbox=tbox, tcls=r["target"][:, 1].long())
            lb, lc, iou = _box_cls_terms(pi, rr, nc, cp, cn)
            lbox = lbox + lb
            _scatter_last_wins(tobj, rr, (1.0 - gr) + gr * iou.detach().clamp(0))
            if lc is not None:
                lcls = lcls + lc
        lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., -1], tobj) * balance[i]      # :246
    asg = _assign(p, anchors, targets, anchor_t)
    for i, pi in enumerate(p):
        r = asg[i]
        tobj = torch.zeros_like(pi[..., 0])
        if r["b"].shape[0]:
            lb, lc, iou = _box_cls_terms(pi, r, nc, cp, cn)
            lbox = lbox + lb
            _scatter_last_wins(tobj, r, (1.0 - gr) + gr * iou.detach().clamp(0))
            if lc is not None:
                lcls = lcls + lc
        lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4], tobj) * balance[i]
    lbox, lobj, lcls = lbox * box_w, lobj * obj_w, lcls * cls_w
    bs = p[0].shape[0]
    loss = lbox + lobj + lcls
    return loss * bs, dict(box=lbox, obj=lobj, cls=lcls, loss=loss * bs)
