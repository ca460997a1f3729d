"""TEST INFRASTRUCTURE -- the whole SSOD step of the reference as one plain-torch CPU function.

Restates SSODTrainer.train_instance (trainer/ssod_trainer.py:587-680): EMA-teacher inference on the weak view
(:595-599), non_max_suppression_ssod + FairPseudoLabel (:618), student forward on cat(labeled, strong view)
(:623-627), ComputeLoss + ComputeStudentMatchLoss (:628-649) and backward (:651), out of the restatements in
oracle/{model,nms,pseudo_label,losses}.py, each of which is pinned against the live reference by
oracle/make_golden.py.  Callers: tests/test_step_fullsize.py (parity of the HIP path at BASELINE.json's own
sizes), bench.py's cpu_baseline / parity_check leg.  Never imported by the product.
"""
import numpy as np
import torch

from . import losses as o_loss, nms as o_nms, pseudo_label as o_pl


def loss_hyper_params(cfg, nl=3):
    """the constructor arithmetic of ComputeLoss (models/loss/loss.py:122-124) and ComputeStudentMatchLoss
    (models/loss/ssod/ssod_loss.py:50-53) for a CfgNode"""
    nc = cfg.Dataset.nc
    sup = dict(nc=nc, box_w=cfg.Loss.box * 3.0 / nl, obj_w=cfg.Loss.obj, cls_w=cfg.Loss.cls * nc / 80. * 3. / nl,
               anchor_t=cfg.Loss.anchor_t)
    uns = dict(nc=nc, box_w=cfg.SSOD.box_loss_weight, obj_w=cfg.SSOD.obj_loss_weight,
               cls_w=cfg.SSOD.cls_loss_weight * nc / 80. * 3. / nl, anchor_t=cfg.Loss.anchor_t,
               thr_low=[cfg.SSOD.ignore_thres_low] * nc, thr_high=[cfg.SSOD.ignore_thres_high] * nc,
               ignore_obj=cfg.SSOD.ignore_obj, with_obj=cfg.SSOD.pseudo_label_with_obj,
               with_bbox=cfg.SSOD.pseudo_label_with_bbox, with_cls=cfg.SSOD.pseudo_label_with_cls)
    return sup, uns


def ssod_step(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg, *, synth_scores=None, teacher_pred=None,
              backward=True):
    """One train_instance up to (and including) backward.  ``synth_scores`` (Bu, A, 1+nc) replaces the teacher's
    objectness / class scores (a random-init teacher detects nothing); ``teacher_pred`` replaces the
    whole decoded teacher output (so that NMS decisions can be compared on bit-identical inputs).
    Returns a dict of everything a parity test wants to look at."""
    n_img = imgs.shape[0]
    height, width = u_str.shape[2], u_str.shape[3]
    with torch.no_grad():
        if teacher_pred is None:
            (tp, _), _ = teacher(u_ori)
            if synth_scores is not None:
                tp[..., 4:] = synth_scores
        else:
            tp = teacher_pred
    dets, keep = o_nms.non_max_suppression_ssod(tp.numpy(), cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    t9, invalid = o_pl.create_pseudo_label(dets, M_s.numpy(), width, height)
    pred, feats = student(torch.cat([imgs, u_str], 0))
    sup = [p[:n_img] for p in pred]
    uns = [p[n_img:] for p in pred]
    hp_s, hp_u = loss_hyper_params(cfg, len(pred))
    sup_loss, sup_items = o_loss.compute_loss(sup, targets, student.head.anchors, **hp_s)
    if not invalid:
        un_loss, un_items = o_loss.compute_student_match_loss(uns, torch.from_numpy(t9), student.head.anchors, **hp_u)
    else:                                                   # ssod_trainer.py:640-643
        un_loss = torch.zeros(1)
        un_items = dict(ss_box=torch.zeros(1), ss_obj=torch.zeros(1), ss_cls=torch.zeros(1))
    loss = sup_loss + un_loss * cfg.SSOD.teacher_loss_weight
    if backward:
        loss.backward()
    return dict(teacher_pred=tp, dets=dets, keep=keep, t9=np.asarray(t9), invalid=invalid, pred=pred, loss=loss,
                sup_items={k: float(v.detach()) for k, v in sup_items.items()},
                un_items={k: float(v.detach()) for k, v in un_items.items()})


def ssod_step_v8(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg, *, synth_scores=None, teacher_pred=None, backward=True):
    """The same step on the anchor-free YOLOv8 head -- an EXTENSION: the reference cannot run it (trainer/ssod_trainer.py:598-606
    raises for model types other than yolov5).  Composition of oracle/v8.py (model, tal_loss, and the written specification
    tal_student_match_loss of the unsupervised term) with the reference-pinned NMS and pseudo-label restatements."""
    from . import v8 as o_v8
    n_img = imgs.shape[0]
    height, width = u_str.shape[2], u_str.shape[3]
    with torch.no_grad():
        if teacher_pred is None:
            tp, _ = teacher(u_ori)
            if synth_scores is not None:
                tp[..., 4:] = synth_scores
        else:
            tp = teacher_pred
    dets, keep = o_nms.non_max_suppression_ssod(tp.numpy(), cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    t9, invalid = o_pl.create_pseudo_label(dets, M_s.numpy(), width, height)
    feats, cls, reg = student(torch.cat([imgs, u_str], 0))
    kw = dict(nc=cfg.Dataset.nc, reg_max=cfg.Loss.reg_max, img_size=cfg.Dataset.img_size, iou_type=cfg.Loss.iou_type,
              w_class=cfg.Loss.qfl_loss_weight, w_iou=cfg.Loss.box_loss_weight, w_dfl=cfg.Loss.dfl_loss_weight)
    sup_loss, sup_items = o_v8.tal_loss(([f[:n_img] for f in feats], cls[:n_img], reg[:n_img]), targets, **kw)
    nc = cfg.Dataset.nc
    if not invalid:
        un_loss, un_items = o_v8.tal_student_match_loss(([f[n_img:] for f in feats], cls[n_img:], reg[n_img:]), np.asarray(t9),
                                                        [cfg.SSOD.ignore_thres_low] * nc, [cfg.SSOD.ignore_thres_high] * nc,
                                                        with_obj=cfg.SSOD.pseudo_label_with_obj, with_bbox=cfg.SSOD.pseudo_label_with_bbox,
                                                        with_cls=cfg.SSOD.pseudo_label_with_cls, **kw)
    else:
        un_loss = torch.zeros(1)
        un_items = dict(ss_box=torch.zeros(1), ss_dfl=torch.zeros(1), ss_cls=torch.zeros(1))
    loss = sup_loss + un_loss * cfg.SSOD.teacher_loss_weight
    if backward:
        loss.backward()
    return dict(teacher_pred=tp, dets=dets, keep=keep, t9=np.asarray(t9), invalid=invalid, loss=loss,
                sup_items={k: float(torch.as_tensor(sup_items[k]).detach()) for k in ("loss_iou", "loss_dfl", "loss_cls")},
                un_items={k: float(torch.as_tensor(un_items[k]).detach()) for k in ("ss_box", "ss_dfl", "ss_cls")})
