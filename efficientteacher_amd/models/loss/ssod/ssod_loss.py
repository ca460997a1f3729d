"""Host-side mirror of the reference's ``models/loss/ssod/ssod_loss.py``.

``ComputeStudentMatchLoss(model, cfg)(p, targets9) -> (loss*bs [1], dict(ss_box, ss_obj, ss_cls))``
with the mutable per-class ``ignore_thres_high / ignore_thres_low`` lists the trainer reads and writes
(trainer/ssod_trainer.py:322-323, 664).  ``select_targets`` (ssod_loss.py:130-192) and the loss
(:194-288) run on the device: there is no per-row ``.cpu()`` and no host synchronisation.
"""
import torch

from .... import ops
from ..loss import YoloLossFn, _head_of, smooth_BCE


class ComputeStudentMatchLoss:
    def __init__(self, model, cfg):
        if cfg.SSOD.focal_loss > 0:
            raise NotImplementedError("SSOD.focal_loss > 0 references an undefined FocalLoss in the reference "
                                      "(ssod_loss.py:41-42)")
        if cfg.Loss.autobalance:
            raise NotImplementedError("Loss.autobalance is off in every shipped config")
        self.cls_pw, self.obj_pw = float(cfg.Loss.cls_pw), float(cfg.Loss.obj_pw)
        self.cp, self.cn = smooth_BCE(eps=cfg.Loss.label_smoothing)
        det = _head_of(model)
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.ssi = 0
        self.gr, self.autobalance = 1.0, False
        self.box_w = cfg.SSOD.box_loss_weight
        self.obj_w = cfg.SSOD.obj_loss_weight
        self.cls_w = cfg.SSOD.cls_loss_weight * cfg.Dataset.nc / 80. * 3. / det.nl
        self.anchor_t = cfg.Loss.anchor_t
        self.ignore_thres_high = [cfg.SSOD.ignore_thres_high] * cfg.Dataset.nc
        self.ignore_thres_low = [cfg.SSOD.ignore_thres_low] * cfg.Dataset.nc
        self.uncertain_aug = cfg.SSOD.uncertain_aug
        self.use_ota = cfg.SSOD.use_ota
        if self.use_ota:
            raise NotImplementedError("SSOD.use_ota raises TypeError in the reference (ssod_loss.py:302-303)")
        self.ignore_obj = cfg.SSOD.ignore_obj
        self.pseudo_label_with_obj = cfg.SSOD.pseudo_label_with_obj
        self.pseudo_label_with_bbox = cfg.SSOD.pseudo_label_with_bbox
        self.pseudo_label_with_cls = cfg.SSOD.pseudo_label_with_cls
        self.num_keypoints = cfg.Dataset.np
        self.single_targets = not self.uncertain_aug
        for k in 'na', 'nc', 'nl', 'anchors', 'stride':
            setattr(self, k, getattr(det, k))
        self._anchors_host = [[[float(v) for v in a] for a in lvl] for lvl in det.anchors.detach().cpu().tolist()]
        self._thr_dev = ops.DeviceThresholds()

    def refresh_thresholds(self, dev):
        """upload ignore_thres_low / _high into their persistent device tensor if the lists changed (LabelMatch's after_epoch
        rewrites them); trainer/graph_step.py calls this before every replay of a captured step"""
        return self._thr_dev.refresh(self.ignore_thres_low, self.ignore_thres_high, dev)

    def select_targets(self, targets, valid=None):
        """(N,9) [batch, cls, x, y, w, h, conf, obj_conf, cls_conf] -> device target table (N,8)."""
        return ops.select_targets(targets, valid, self.ignore_thres_low, self.ignore_thres_high, self.nc,
                                  self.pseudo_label_with_obj, thresholds=self._thr_dev)

    def _hp(self, pass_mask):
        return dict(nc=self.nc, anchor_t=float(self.anchor_t), gr=float(self.gr), cp=float(self.cp), cn=float(self.cn),
                    cls_pw=self.cls_pw, obj_pw=self.obj_pw, box_w=float(self.box_w), obj_w=float(self.obj_w),
                    cls_w=float(self.cls_w), pass_mask=pass_mask, ignore_obj=bool(self.ignore_obj))

    def default_loss(self, p, targets, valid=None):
        dev = p[0].device
        targets = targets.to(dev)
        if targets.shape[1] > 6:
            table = self.select_targets(targets, valid)
            mask = 1 | 2 | (4 if self.pseudo_label_with_bbox else 0) | (8 if self.pseudo_label_with_cls else 0)
        else:  # plain (n,6) targets: same as the supervised loss (ssod_loss.py:208-209)
            t = targets[:, :6].float()
            n = t.shape[0]
            table = torch.cat((t, torch.zeros((n, 1), device=dev), torch.ones((n, 1), device=dev)), 1)
            mask = 1
        hp = self._hp(mask)
        hp["grad_dst"] = [getattr(pi, "_et_grad_dst", None) for pi in p]
        out = YoloLossFn.apply(table, hp, self._anchors_host, self.balance, *p)
        det = out.detach()
        return out[3:4], dict(ss_box=det[0:1], ss_obj=det[1:2], ss_cls=det[2:3])

    def __call__(self, p, targets, valid=None):
        return self.default_loss(p, targets, valid)
