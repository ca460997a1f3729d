"""TEST INFRASTRUCTURE -- plain-torch / numpy restatement of the reference's YOLOv8 anchor-free path.

  C2f                      models/backbone/common.py:594-608
  Backbone / Neck          models/backbone/yolov8_backbone.py:25-100, models/neck/yolov8_neck.py:6-118
  Detect (train + eval)    models/head/yolov8_head.py:10-214, generate_anchors / dist2bbox models/module/nanodet_utils.py:92-180
  tal_assign               models/assigner/tal_assigner.py:13-158 + select_candidates_in_gts, select_highest_overlaps,
                           iou_calculator (nanodet_utils.py:181-243)
  tal_loss                 models/loss/tal_loss.py:16-156.  PARITY UNPINNED for this one function: it imports
                           models.loss.gfocal_loss.{VarifocalLoss, BboxLoss} and models.assigner.yolo_atss_assigner, which are
                           ABSENT from the reference tree, so ComputeTalLoss cannot be imported or run.  The
                           restatement follows tal_loss.py line by line and fills the two missing classes with their upstream
                           definitions (meituan/YOLOv6 yolov6/models/losses/loss.py: BboxLoss = IoU loss weighted by the
                           target score + DFL cross-entropy on bbox2dist targets; VarifocalLoss alpha 0.75 gamma 2.0).
Everything else is pinned against the LIVE reference by oracle/make_golden.py::case_v8 (tests/golden/v8_*.npz).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import SPPF, Conv, make_divisible


class Bottleneck33(nn.Module):
    """Bottleneck(c, c, shortcut, k=(3, 3), e=1.0) as C2f builds it"""

    def __init__(self, c1, c2, shortcut=True, e=1.0):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 3, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C2f(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck33(self.c, self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).split((self.c, self.c), 1))
        y.extend(m(y[-1]) for m in self.m)
        return self.cv2(torch.cat(y, 1))


class Backbone(nn.Module):
    def __init__(self, gw, gd):
        super().__init__()
        w = lambda n: make_divisible(n * gw, 8)
        d = lambda n: max(round(n * gd), 1) if n > 1 else n
        self.stage1 = Conv(3, w(64), 6, 2, 2)
        self.stage2_1 = Conv(w(64), w(128), 3, 2)
        self.stage2_2 = C2f(w(128), w(128), d(3), True)
        self.stage3_1 = Conv(w(128), w(256), 3, 2)
        self.stage3_2 = C2f(w(256), w(256), d(6), True)
        self.stage4_1 = Conv(w(256), w(512), 3, 2)
        self.stage4_2 = C2f(w(512), w(512), d(6), True)
        self.stage5_1 = Conv(w(512), w(768), 3, 2)
        self.stage5_2 = C2f(w(768), w(768), d(3), True)
        self.sppf = SPPF(w(768), w(768), 5)

    def forward(self, x):
        x22 = self.stage2_2(self.stage2_1(self.stage1(x)))
        c3 = self.stage3_2(self.stage3_1(x22))
        c4 = self.stage4_2(self.stage4_1(c3))
        return c3, c4, self.sppf(self.stage5_2(self.stage5_1(c4)))


class Neck(nn.Module):
    def __init__(self, gw, gd, cin=(256, 512, 768), cout=(256, 512, 768)):
        super().__init__()
        w = lambda n: make_divisible(n * gw, 8)
        d = lambda n: max(round(n * gd), 1) if n > 1 else n
        i3, i4, i5 = (w(c) for c in cin)
        o3, o4, o5 = (w(c) for c in cout)
        self.C1 = C2f(i5 + i4, i4, d(3), False)
        self.C2 = C2f(i4 + i3, o3, d(3), False)
        self.conv3 = Conv(o3, o3, 3, 2)
        self.C3 = C2f(o3 + i4, o4, d(3), False)
        self.conv4 = Conv(o4, o4, 3, 2)
        self.C4 = C2f(o4 + i5, o5, d(3), False)

    def forward(self, inputs):
        P3, P4, P5 = inputs
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        x1 = self.C1(torch.cat([up(P5), P4], 1))
        x2 = self.C2(torch.cat([up(x1), P3], 1))
        x3 = self.C3(torch.cat([self.conv3(x2), x1], 1))
        x4 = self.C4(torch.cat([self.conv4(x3), P5], 1))
        return [x2, x3, x4]


def anchor_points_eval(shapes, strides, offset=0.5):
    """generate_anchors(..., is_eval=True) (nanodet_utils.py:132-147): cell centres in GRID units + per-anchor stride"""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sy, sx = torch.meshgrid(torch.arange(h) + offset, torch.arange(w) + offset, indexing="ij")
        pts.append(torch.stack([sx, sy], -1).float().reshape(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def anchor_points_train(shapes, strides, offset=0.5):
    """generate_anchors(..., is_eval=False) (:148-180): cell centres in PIXELS + per-anchor stride"""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sy, sx = torch.meshgrid((torch.arange(h) + offset) * s, (torch.arange(w) + offset) * s, indexing="ij")
        pts.append(torch.stack([sx, sy], -1).float().reshape(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, box_format="xyxy"):
    lt, rb = torch.split(distance, 2, -1)
    x1y1, x2y2 = anchor_points - lt, anchor_points + rb
    if box_format == "xyxy":
        return torch.cat([x1y1, x2y2], -1)
    return torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1)


class Detect(nn.Module):
    def __init__(self, nc, ch, reg_max=16, strides=(8, 16, 32)):
        super().__init__()
        self.nc, self.reg_max, self.nl = nc, reg_max, len(ch)
        self.stride = torch.tensor(strides).float()
        self.proj_conv = nn.Conv2d(reg_max + 1, 1, 1, bias=False)
        c2, c3 = max((16, ch[0] // 4, (reg_max + 1) * 4)), max(ch[0], nc)
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3, 1), Conv(c2, c2, 3, 1), nn.Conv2d(c2, 4 * (reg_max + 1), 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3, 1), Conv(c3, c3, 3, 1), nn.Conv2d(c3, nc, 1)) for x in ch)
        for a, b, s in zip(self.cv2, self.cv3, self.stride):                # initialize_biases (:80-86)
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[:nc] = math.log(5 / nc / (640 / s) ** 2)
        self.proj = nn.Parameter(torch.linspace(0, reg_max, reg_max + 1), requires_grad=False)
        self.proj_conv.weight = nn.Parameter(self.proj.view(1, reg_max + 1, 1, 1).clone(), requires_grad=False)

    def forward(self, x):
        cls, reg = [], []
        for i in range(self.nl):
            reg.append(self.cv2[i](x[i]).flatten(2).permute(0, 2, 1))
            cls.append(self.cv3[i](x[i]).flatten(2).permute(0, 2, 1))
        cls, reg = torch.cat(cls, 1), torch.cat(reg, 1)
        if self.training:
            return x, cls, reg
        pts, st = anchor_points_eval([t.shape[-2:] for t in x], self.stride)
        B, A, _ = reg.shape
        dist = F.softmax(reg.view(B, A, 4, self.reg_max + 1), -1).matmul(self.proj)        # == proj_conv on the softmax (:196-198)
        box = dist2bbox(dist, pts, "xywh") * st
        return torch.cat([box, torch.ones(B, A, 1), cls.sigmoid()], -1), (x, cls, reg)


class Model(nn.Module):
    def __init__(self, gw=0.75, gd=0.67, nc=80, reg_max=16, neck_ch=(256, 512, 768)):
        super().__init__()
        self.backbone = Backbone(gw, gd)
        self.neck = Neck(gw, gd, neck_ch, neck_ch)
        self.head = Detect(nc, [int(c * gw) for c in neck_ch], reg_max)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    @classmethod
    def from_cfg(cls, cfg):
        return cls(cfg.Model.width_multiple, cfg.Model.depth_multiple, cfg.Dataset.nc, cfg.Loss.reg_max, tuple(cfg.Model.Neck.out_channels))

    def forward(self, x):
        return self.head(self.neck(self.backbone(x)))


# ---- TaskAlignedAssigner -----------------------------------------------------------------------------------------------
def iou_calculator(box1, box2, eps=1e-9):
    box1, box2 = box1.unsqueeze(2), box2.unsqueeze(1)
    x1y1 = torch.maximum(box1[..., 0:2], box2[..., 0:2])
    x2y2 = torch.minimum(box1[..., 2:4], box2[..., 2:4])
    overlap = (x2y2 - x1y1).clip(0).prod(-1)
    area1 = (box1[..., 2:4] - box1[..., 0:2]).clip(0).prod(-1)
    area2 = (box2[..., 2:4] - box2[..., 0:2]).clip(0).prod(-1)
    return overlap / (area1 + area2 - overlap + eps)


def tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk=13, alpha=1.0, beta=6.0, eps=1e-9, return_idx=False):
    """tal_assigner.py:30-158.  Ties between EQUAL metrics go to the smaller anchor index (torch.topk leaves them
    unspecified; see csrc/tal.hip) -- implemented with a stable descending sort."""
    bs, A, nc = pd_scores.shape
    G = gt_bboxes.shape[1]
    if G == 0:
        r = (torch.full((bs, A), nc, dtype=torch.int64), torch.zeros(bs, A, 4), torch.zeros(bs, A, nc), torch.zeros(bs, A, dtype=torch.bool))
        return r + (torch.zeros(bs, A, dtype=torch.int64),) if return_idx else r
    lab = gt_labels.long().squeeze(-1)
    bbox_scores = pd_scores.permute(0, 2, 1)[torch.arange(bs).view(-1, 1).expand(bs, G), lab]          # (bs, G, A)
    overlaps = iou_calculator(gt_bboxes, pd_bboxes, eps)
    align_metric = bbox_scores.pow(alpha) * overlaps.pow(beta)
    lt = anc_points.view(1, 1, A, 2) - gt_bboxes[..., None, 0:2]
    rb = gt_bboxes[..., None, 2:4] - anc_points.view(1, 1, A, 2)
    mask_in_gts = (torch.cat([lt, rb], -1).min(-1)[0] > eps).float()
    metrics = align_metric * mask_in_gts
    order = torch.sort(metrics, dim=-1, descending=True, stable=True)[1][..., :topk]
    is_in_topk = torch.zeros_like(metrics).scatter_(-1, order, 1.0) * (mask_gt.view(bs, G, 1) > 0).float()
    mask_pos = is_in_topk * mask_in_gts * mask_gt.view(bs, G, 1)
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(bs, G, A)
        is_max = F.one_hot(overlaps.argmax(1), G).permute(0, 2, 1).float()
        mask_pos = torch.where(multi, is_max, mask_pos)
        fg = mask_pos.sum(-2)
    idx = mask_pos.argmax(-2)
    flat = idx + torch.arange(bs).view(-1, 1) * G
    tl = gt_labels.long().flatten()[flat]
    tb = gt_bboxes.reshape(-1, 4)[flat]
    tl = tl.clamp(min=0)
    ts = F.one_hot(tl, nc).float() * (fg > 0).unsqueeze(-1).float()
    am = align_metric * mask_pos
    pos_am = am.max(-1, keepdim=True)[0]
    pos_ov = (overlaps * mask_pos).max(-1, keepdim=True)[0]
    norm = (am * pos_ov / (pos_am + eps)).max(-2)[0].unsqueeze(-1)
    if return_idx:
        return tl, tb, ts * norm, fg > 0, idx
    return tl, tb, ts * norm, fg > 0


# ---- ComputeTalLoss (written spec, parity unpinned: see the module docstring) ------------------------------------------
def bbox_iou_xyxy(b1, b2, kind="giou", eps=1e-7):
    """IoU family on xyxy boxes, as YOLOv6's IOUloss (upstream of the absent models/loss/gfocal_loss.py): returns the IoU term
    whose complement is the loss (iou / giou / ciou)."""
    x1, y1, x2, y2 = b1.unbind(-1)
    X1, Y1, X2, Y2 = b2.unbind(-1)
    w1, h1, w2, h2 = x2 - x1, y2 - y1 + eps, X2 - X1, Y2 - Y1 + eps
    inter = (torch.min(x2, X2) - torch.max(x1, X1)).clamp(0) * (torch.min(y2, Y2) - torch.max(y1, Y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if kind == "iou":
        return iou
    cw, ch = torch.max(x2, X2) - torch.min(x1, X1), torch.max(y2, Y2) - torch.min(y1, Y1)
    if kind == "giou":
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        a = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * a)


def _tal_terms(pred_scores, pred_distri, pred_bboxes, anchor_points_s, stride_tensor, tl, tb, ts, fg, nc, reg_max, use_dfl, use_gfl, iou_type):
    """the three loss terms of ComputeTalLoss for given assigner outputs (tb in pixels); shared by tal_loss and
    tal_student_match_loss"""
    B, A = pred_scores.shape[:2]
    tb = tb / stride_tensor
    if use_gfl:      # VarifocalLoss(alpha 0.75, gamma 2.0)
        label = F.one_hot(torch.where(fg, tl, torch.full_like(tl, nc)), nc + 1)[..., :-1].float()
        weight = 0.75 * pred_scores.sigmoid().pow(2.0) * (1 - label) + ts * label
        loss_cls = (F.binary_cross_entropy_with_logits(pred_scores, ts, reduction="none") * weight).sum()
    else:
        loss_cls = F.binary_cross_entropy_with_logits(pred_scores, ts, reduction="none").sum()
    ts_sum = max(ts.sum(), 1)
    loss_cls = loss_cls / ts_sum
    # BboxLoss: IoU term weighted by the target score, DFL on the distance targets
    if fg.sum() > 0:
        w = ts.sum(-1)[fg].unsqueeze(-1)
        iou = bbox_iou_xyxy(pred_bboxes[fg], tb[fg], iou_type)
        loss_iou = ((1.0 - iou).unsqueeze(-1) * w).sum() / ts_sum
        if use_dfl:
            lt, rb = anchor_points_s.expand(B, A, 2)[fg] - tb[fg][:, :2], tb[fg][:, 2:] - anchor_points_s.expand(B, A, 2)[fg]
            tdist = torch.cat([lt, rb], -1).clip(0, reg_max - 0.01)
            pd = pred_distri.view(B, A, 4, reg_max + 1)[fg].view(-1, reg_max + 1)
            tl_, tr_ = tdist.long(), tdist.long() + 1
            wl, wr = tr_.float() - tdist, tdist - tl_.float()
            ce = (F.cross_entropy(pd, tl_.view(-1), reduction="none").view(tl_.shape) * wl +
                  F.cross_entropy(pd, tr_.view(-1), reduction="none").view(tl_.shape) * wr).mean(-1, keepdim=True)
            loss_dfl = (ce * w).sum() / ts_sum
        else:
            loss_dfl = pred_distri.sum() * 0.0
    else:
        loss_iou, loss_dfl = pred_distri.sum() * 0.0, pred_distri.sum() * 0.0
    return loss_cls, loss_iou, loss_dfl


def tal_loss(outputs, targets, strides=(8, 16, 32), nc=80, reg_max=16, img_size=640, use_dfl=True, use_gfl=False, iou_type="giou",
             w_class=0.5, w_iou=7.5, w_dfl=1.5, cell_offset=0.5):
    """tal_loss.py:51-146.  outputs = (feats, pred_scores (B,A,nc) logits, pred_distri (B,A,4*(reg_max+1))); targets (n,6)
    [img, cls, x, y, w, h] normalised.  Returns (loss [1], dict(loss_iou, loss_dfl, loss_cls, loss, num_fg))."""
    feats, pred_scores, pred_distri = outputs
    pred_scores, pred_distri = pred_scores.float(), pred_distri.float()
    anchor_points, stride_tensor = anchor_points_train([f.shape[-2:] for f in feats], strides, cell_offset)
    B, A = pred_scores.shape[:2]
    # preprocess (:131-143): pad per image, scale to pixels, xywh -> xyxy
    per = [[] for _ in range(B)]
    for row in targets.tolist():
        per[int(row[0])].append(row[1:])
    G = max(max(len(p) for p in per), 0)
    num_gts = sum(len(p) for p in per) + B            # the reference counts its one dummy row per image (:133-137)
    tt = torch.zeros(B, G, 5)
    tt[..., 0] = -1
    for i, p in enumerate(per):
        if p:
            tt[i, :len(p)] = torch.tensor(p)
    box = tt[..., 1:5] * img_size
    gt_bboxes = torch.stack([box[..., 0] - box[..., 2] * 0.5, box[..., 1] - box[..., 3] * 0.5,
                             box[..., 0] - box[..., 2] * 0.5 + box[..., 2], box[..., 1] - box[..., 3] * 0.5 + box[..., 3]], -1)
    gt_labels = tt[..., :1]
    mask_gt = (gt_bboxes.sum(-1, keepdim=True) > 0).float()
    anchor_points_s = anchor_points / stride_tensor
    proj = torch.linspace(0, reg_max, reg_max + 1)
    dist = F.softmax(pred_distri.view(B, A, 4, reg_max + 1), -1).matmul(proj) if use_dfl else pred_distri
    pred_bboxes = dist2bbox(dist, anchor_points_s)
    tl, tb, ts, fg = tal_assign(pred_scores.detach().sigmoid(), pred_bboxes.detach() * stride_tensor, anchor_points, gt_labels,
                                gt_bboxes, mask_gt)
    loss_cls, loss_iou, loss_dfl = _tal_terms(pred_scores, pred_distri, pred_bboxes, anchor_points_s, stride_tensor, tl, tb, ts, fg, nc,
                                              reg_max, use_dfl, use_gfl, iou_type)
    loss = torch.zeros(1) + w_class * loss_cls + w_iou * loss_iou + w_dfl * loss_dfl
    return loss, dict(loss_iou=w_iou * loss_iou, loss_dfl=w_dfl * loss_dfl, loss_cls=w_class * loss_cls, loss=loss,
                      num_fg=fg.sum() / max(num_gts, 1))


# ---- EXTENSION: ComputeStudentMatchLoss on the anchor-free head ---------------------------------------------------------------
def tal_student_match_loss(outputs, targets9, thr_low, thr_high, strides=(8, 16, 32), nc=80, reg_max=16, img_size=640,
                           iou_type="giou", w_class=0.5, w_iou=7.5, w_dfl=1.5, cell_offset=0.5, with_obj=True, with_bbox=True,
                           with_cls=False):
    """SPECIFICATION of the TAL variant of ComputeStudentMatchLoss.  **Not in the reference** (its ComputeStudentMatchLoss needs
    det.anchors, models/loss/ssod/ssod_loss.py:69, and trainer/ssod_trainer.py:598-606 raises for model types other than yolov5;
    update_train_logger :271-272 merely anticipates a 'tal' variant): parity is UNPINNED BY CONSTRUCTION, this function IS the
    definition that csrc/tal.hip (et_tal_pseudo_split / et_tal_merge_pseudo + et_tal_assign / et_tal_loss) is tested against.

    It carries ssod_loss.py:130-296 over to TaskAlignedAssigner targets.  targets9 (N, 9) = [img, cls, x, y, w, h (normalised),
    conf, obj_conf, cls_conf]:
      split (:130-192)   reliable  R = {conf >= thr_high[cls]};  uncertain U = {thr_low[cls] <= conf < thr_high[cls]} with the soft
                         score s = obj_conf (with_obj) else conf;  U_box = {u in U: obj_conf >= 0.99},  U_cls = {u in U: cls_conf >= 0.99}
                         (both formed only under with_obj, as in the reference)
      assign             R and U are assigned SEPARATELY by the TaskAlignedAssigner on the (detached) student predictions, like the
                         reference runs its anchor assigner once per subset (:199-207)
      merge (:231,:248)  an anchor owned by an uncertain label takes that label's targets even if a reliable label owns it too (the
                         reference writes tobj for the reliable cells first and for the uncertain cells afterwards);
                         class target of an uncertain anchor = aligned score * s  (the soft objectness target of :248 on a head whose
                         class score IS its objectness), or the un-scaled aligned score for U_cls under with_cls (:268-277);
                         box / DFL terms: reliable anchors, and uncertain anchors of U_box under with_bbox (:251-266)
      loss               the three ComputeTalLoss terms on the merged targets, same normalisation (sum of target scores) and weights.
    Returns (loss [1], dict(ss_box, ss_dfl, ss_cls))."""
    feats, pred_scores, pred_distri = outputs
    pred_scores, pred_distri = pred_scores.float(), pred_distri.float()
    anchor_points, stride_tensor = anchor_points_train([f.shape[-2:] for f in feats], strides, cell_offset)
    B, A = pred_scores.shape[:2]
    rows = [[float(v) for v in r] for r in (targets9.tolist() if hasattr(targets9, "tolist") else targets9)]
    rel = [[] for _ in range(B)]
    unc = [[] for _ in range(B)]
    for r in rows:
        b, c = int(r[0]), min(max(int(r[1]), 0), nc - 1)
        x1, y1 = np.float32(r[2]) * np.float32(img_size) - np.float32(r[4]) * np.float32(img_size) * np.float32(0.5), \
            np.float32(r[3]) * np.float32(img_size) - np.float32(r[5]) * np.float32(img_size) * np.float32(0.5)
        box = [x1, y1, x1 + np.float32(r[4]) * np.float32(img_size), y1 + np.float32(r[5]) * np.float32(img_size)]
        if r[6] >= thr_high[c]:
            rel[b].append([c] + box)
        elif r[6] >= thr_low[c]:
            s = r[7] if with_obj else r[6]
            unc[b].append([c] + box + [s, 1.0 if (with_obj and with_bbox and r[7] >= 0.99) else 0.0,
                                       1.0 if (with_obj and with_cls and r[8] >= 0.99) else 0.0])

    def padded(per, width):
        G = max(max(len(p) for p in per), 1)
        t = torch.zeros(B, G, width)
        t[..., 0] = -1
        m = torch.zeros(B, G, 1)
        for i, p in enumerate(per):
            if p:
                t[i, :len(p)] = torch.tensor(p, dtype=torch.float32)
                m[i, :len(p)] = 1
        return t, m
    tr, mr = padded(rel, 5)
    tu, mu = padded(unc, 8)
    anchor_points_s = anchor_points / stride_tensor
    proj = torch.linspace(0, reg_max, reg_max + 1)
    dist = F.softmax(pred_distri.view(B, A, 4, reg_max + 1), -1).matmul(proj)
    pred_bboxes = dist2bbox(dist, anchor_points_s)
    sc, bx = pred_scores.detach().sigmoid(), pred_bboxes.detach() * stride_tensor
    tl_r, tb_r, ts_r, fg_r = tal_assign(sc, bx, anchor_points, tr[..., :1], tr[..., 1:5], mr)
    tl_u, tb_u, ts_u, fg_u, idx_u = tal_assign(sc, bx, anchor_points, tu[..., :1], tu[..., 1:5], mu, return_idx=True)
    bi = torch.arange(B).view(-1, 1).expand(B, A)
    s_u, box_u, cls_u = tu[bi, idx_u, 5], tu[bi, idx_u, 6] > 0, tu[bi, idx_u, 7] > 0
    scale_u = torch.where(cls_u, torch.ones_like(s_u), s_u)
    ts = torch.where(fg_u.unsqueeze(-1), ts_u * scale_u.unsqueeze(-1), torch.where(fg_r.unsqueeze(-1), ts_r, torch.zeros_like(ts_r)))
    tb = torch.where(fg_u.unsqueeze(-1), tb_u, tb_r)
    tl = torch.where(fg_u, tl_u, tl_r)
    fg_box = (fg_u & box_u) | (fg_r & ~fg_u)
    loss_cls, loss_iou, loss_dfl = _tal_terms(pred_scores, pred_distri, pred_bboxes, anchor_points_s, stride_tensor, tl, tb, ts, fg_box, nc,
                                              reg_max, True, False, iou_type)
    loss = torch.zeros(1) + w_class * loss_cls + w_iou * loss_iou + w_dfl * loss_dfl
    return loss, dict(ss_box=w_iou * loss_iou, ss_dfl=w_dfl * loss_dfl, ss_cls=w_class * loss_cls, n_reliable=sum(len(p) for p in rel),
                      n_uncertain=sum(len(p) for p in unc), n_fg_box=int(fg_box.sum()), n_fg_soft=int(fg_u.sum()))
