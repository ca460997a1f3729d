"""Host-side mirror of the reference's ``utils/self_supervised_utils.py`` for the hot path:
``FairPseudoLabel`` (:54) turns the EMA teacher's decoded predictions into pseudo labels.

``create_pseudo_label_online_with_gt`` keeps the reference signature / return value (a compacted
(N,9) float64 tensor and ``invalid_target_shape``), which costs one host synchronisation;
``create_pseudo_label_padded`` is the device-resident form the trainer uses (no synchronisation).
"""
from .. import ops
from .general import nms_ssod_padded


class FairPseudoLabel:
    def __init__(self, cfg):
        self.nms_conf_thres = cfg.SSOD.nms_conf_thres
        self.nms_iou_thres = cfg.SSOD.nms_iou_thres
        self.debug = cfg.SSOD.debug
        self.multi_label = cfg.SSOD.multi_label
        self.names = cfg.Dataset.names
        self.num_points = cfg.Dataset.np
        if self.multi_label or self.num_points:
            raise NotImplementedError("multi_label / keypoint pseudo labels are outside the hot path")

    def create_pseudo_label_padded(self, out, M_s, width, height, max_det=300):
        """out (B, A, 5+nc) teacher predictions -> (targets9 (B*max_det, 9) fp64, valid (B*max_det) uint8)."""
        dets, counts, _, _ = nms_ssod_padded(out, self.nms_conf_thres, self.nms_iou_thres, max_det=max_det)
        return ops.pseudo_label_transform(dets, counts, M_s, width, height)

    def create_pseudo_label_online_with_gt(self, out, target_imgs, M_s, target_imgs_ori, gt=None, RANK=-2):
        n_img, _, height, width = target_imgs.shape
        t9, valid = self.create_pseudo_label_padded(out, M_s, width, height)
        targets = t9[valid.bool()]                  # compaction: the one host sync of this API
        invalid_target_shape = targets.shape[0] == 0
        return targets, invalid_target_shape
