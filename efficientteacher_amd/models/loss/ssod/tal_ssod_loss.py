"""EXTENSION -- ``ComputeStudentMatchLoss`` for the anchor-free YOLOv8 head (BASELINE.json configs[4] is an SSOD config).

The reference has NO such class: its ``ComputeStudentMatchLoss`` reads ``det.anchors`` (models/loss/ssod/ssod_loss.py:69) and its
SSOD trainer raises for model types other than yolov5 (trainer/ssod_trainer.py:598-606); ``update_train_logger`` (:271-272)
merely anticipates a ``'tal'`` variant.  This class carries the reliable / uncertain pseudo-label logic of ssod_loss.py:130-296
over to TaskAlignedAssigner targets.  Its definition is the written specification ``oracle/v8.py::tal_student_match_loss``
(parity UNPINNED by construction -- there is nothing in the reference to pin it on); tests/test_v8.py checks the kernels against it.

``ComputeStudentMatchTalLoss(model, cfg)(outputs, targets9[, valid]) -> (loss [1], dict(ss_box, ss_dfl, ss_cls))`` with the mutable
per-class ``ignore_thres_high / ignore_thres_low`` lists of the anchor-based class (LabelMatch rewrites them).  Everything runs
on the device: et_tal_pseudo_split -> et_tal_assign (reliable) + et_tal_assign (uncertain) -> et_tal_merge_pseudo -> et_tal_loss.
"""
import torch

from .... import ops
from ..tal_loss import ComputeTalLoss


class ComputeStudentMatchTalLoss:
    def __init__(self, model, cfg):
        if cfg.SSOD.ignore_obj:
            raise NotImplementedError("SSOD.ignore_obj on the anchor-free head: there is no objectness cell to ignore")
        if cfg.SSOD.use_ota:
            raise NotImplementedError("SSOD.use_ota raises TypeError in the reference (ssod_loss.py:302-303)")
        self.tal = ComputeTalLoss(model, cfg)          # decode / anchors / loss weights of the supervised TAL loss
        self.nc = cfg.Dataset.nc
        self.img_size = cfg.Dataset.img_size
        self.ignore_thres_high = [cfg.SSOD.ignore_thres_high] * cfg.Dataset.nc
        self.ignore_thres_low = [cfg.SSOD.ignore_thres_low] * cfg.Dataset.nc
        self.pseudo_label_with_obj = cfg.SSOD.pseudo_label_with_obj
        self.pseudo_label_with_bbox = cfg.SSOD.pseudo_label_with_bbox
        self.pseudo_label_with_cls = cfg.SSOD.pseudo_label_with_cls
        self._thr_dev = ops.DeviceThresholds()

    def refresh_thresholds(self, dev):
        return self._thr_dev.refresh(self.ignore_thres_low, self.ignore_thres_high, dev)

    @staticmethod
    def _padded(targets9, B, dev):
        """compacted (N, 9) rows (the reference's calling convention) -> per-image padded table + valid mask (host side, one sync)"""
        t = targets9.detach().to("cpu", torch.float64)
        per = [[] for _ in range(B)]
        for row in t.tolist():
            per[int(row[0])].append(row)
        G = max(max(len(p) for p in per), 1)
        out = torch.zeros(B, G, 9, dtype=torch.float64)
        valid = torch.zeros(B, G, dtype=torch.uint8)
        for i, p in enumerate(per):
            if p:
                out[i, :len(p)] = torch.tensor(p, dtype=torch.float64)
                valid[i, :len(p)] = 1
        return out.view(B * G, 9).to(dev), valid.view(-1).to(dev)

    def __call__(self, outputs, targets9, valid=None):
        feats, pred_scores, pred_distri = outputs
        dev = pred_scores.device
        B, A, nc = pred_scores.shape
        if valid is None or targets9.shape[0] % B:
            targets9, valid = self._padded(targets9, B, dev)
        t9 = targets9.to(device=dev, dtype=torch.float64).contiguous()
        thr = self.refresh_thresholds(dev)
        scores, pd_xyxy, anchor_points, anchor_points_s, stride_tensor = self.tal.assigner_inputs(feats, pred_scores, pred_distri)
        with torch.no_grad():
            rel, unc, us, uf = ops.tal_pseudo_split(t9, valid, thr, B, nc, self.pseudo_label_with_obj, self.pseudo_label_with_bbox,
                                                    self.pseudo_label_with_cls, self.img_size, self.img_size)
            _, tb_r, ts_r, fg_r = ops.tal_assign(scores, pd_xyxy, anchor_points, *rel)
            _, tb_u, ts_u, fg_u, idx_u = ops.tal_assign(scores, pd_xyxy, anchor_points, *unc, return_idx=True)
            tb, ts, fg_box = ops.tal_merge_pseudo((tb_r, ts_r, fg_r), (tb_u, ts_u, fg_u, idx_u), us, uf)
        out = self.tal.loss_terms(pred_scores, pred_distri, anchor_points_s, stride_tensor, tb, ts, fg_box)
        d = out.detach()
        return out[3:4], dict(ss_box=d[0:1], ss_dfl=d[1:2], ss_cls=d[2:3])
