"""Host-side mirror of the reference's ``utils/labelmatch.py`` (``LabelMatch``, :56): pseudo labels with per-class
thresholds that are re-estimated at the end of every epoch from the scores the teacher produced during it.

In the step (``create_pseudo_label_padded``): NMS -> append every detection's (class, confidence) to a device log
(the reference's ``score_list_epoch`` python lists, :279-287) -> the same warp/filter as FairPseudoLabel plus the clip of
:333 -- no host synchronisation.  At the end of the epoch (``update_epoch_cls_thr``, :188-240): the log is read once, and
per class the low threshold is the score at a fixed rank, the high threshold comes from a two-component Gaussian mixture
(``gmm_policy``, :134-186; sklearn.mixture.GaussianMixture as in the reference).
"""
import logging

import numpy as np
import torch

from .. import ops
from .general import nms_ssod_padded

LOGGER = logging.getLogger(__name__)


class LabelMatch:
    def __init__(self, cfg, target_data_len, label_num_per_img, cls_ratio_gt, score_log_capacity=None):
        self.nc = int(np.asarray(cls_ratio_gt).shape[0])
        self.multi_label = cfg.SSOD.multi_label
        self.nms_conf_thres = cfg.SSOD.nms_conf_thres
        self.nms_iou_thres = cfg.SSOD.nms_iou_thres
        self.cls_thr_high = [cfg.SSOD.ignore_thres_high] * self.nc
        self.cls_thr_low = [cfg.SSOD.ignore_thres_low] * self.nc
        self.cls_ratio_gt = cls_ratio_gt
        self.ignore_thres_low = cfg.SSOD.ignore_thres_low
        self.ignore_thres_high = cfg.SSOD.ignore_thres_high
        self.resample_high_percent = cfg.SSOD.resample_high_percent
        self.resample_low_percent = cfg.SSOD.resample_low_percent
        self.debug = cfg.SSOD.debug
        self.names = cfg.Dataset.names
        self.num_points = cfg.Dataset.np
        if self.multi_label or self.num_points:
            raise NotImplementedError("multi_label / keypoint pseudo labels are outside the hot path")
        self.target_data_len = target_data_len
        self.anno_num_per_img = label_num_per_img * 3
        self.cls_num_total = np.zeros(self.nc)
        self.cls_tmp = np.zeros(self.nc)
        self.count = 0
        self.pse_count = 0
        # device log of one epoch's (class, confidence) pairs; 64 detections per unlabeled image on average is ~3x what
        # conf_thres 0.1 lets through on COCO.  An overflow is an error at the end of the epoch, never a silent truncation.
        self.capacity = int(score_log_capacity or max(1 << 16, 64 * int(target_data_len)))
        self._log = None
        self._cls_hist = None

    # -- per step -------------------------------------------------------------------------------------------------
    def _ensure_log(self, dev):
        if self._log is None:
            self._log = (torch.empty(self.capacity, dtype=torch.float32, device=dev),
                         torch.empty(self.capacity, dtype=torch.int32, device=dev),
                         torch.zeros(1, dtype=torch.int64, device=dev))
        return self._log

    def update(self, labels, n=1, pse_n=1):
        """:122-132 class histogram of the labelled targets seen this epoch (bookkeeping only; no host sync)"""
        self.count += n
        self.pse_count += pse_n
        if labels is None or labels.numel() == 0:
            return
        idx = labels[:, 1].to(torch.int64).clamp_(0, self.nc - 1)
        if self._cls_hist is None or self._cls_hist.device != idx.device:
            self._cls_hist = torch.zeros(self.nc, dtype=torch.float32, device=idx.device)
        # index_add_, not bincount: bincount sizes its output from the data (a device->host synchronisation per step)
        self._cls_hist.index_add_(0, idx, torch.ones(idx.shape[0], dtype=torch.float32, device=idx.device))

    def create_pseudo_label_padded(self, out, M_s, width, height, max_det=300):
        """out (B, A, 5+nc) teacher predictions -> (targets9 (B*max_det, 9) fp64, valid (B*max_det) uint8)."""
        dets, counts, _, _ = nms_ssod_padded(out, self.nms_conf_thres, self.nms_iou_thres, max_det=max_det)
        conf_log, cls_log, n_log = self._ensure_log(dets.device)
        ops.score_log_append(dets, counts, conf_log, cls_log, n_log)
        return ops.pseudo_label_transform(dets, counts, M_s, width, height, clip01=True)

    def create_pseudo_label_online_with_gt(self, out, target_imgs, M_s, target_imgs_ori, gt=None, RANK=-2):
        """reference signature (:271): compacted (N,9) fp64 targets and invalid_target_shape -- one host sync"""
        n_img, _, height, width = target_imgs.shape
        t9, valid = self.create_pseudo_label_padded(out, M_s, width, height)
        targets = t9[valid.bool()]
        return targets, targets.shape[0] == 0

    # -- per epoch ------------------------------------------------------------------------------------------------
    def epoch_scores(self):
        """per-class score lists of the running epoch, sorted descending (host; one D2H copy)"""
        lists = [[] for _ in range(self.nc)]
        if self._log is None:
            return lists
        conf_log, cls_log, n_log = self._log
        n = int(n_log.item())
        if n > self.capacity:
            raise RuntimeError(f"LabelMatch score log overflow: {n} detections this epoch, capacity {self.capacity}; "
                               "construct LabelMatch with a larger score_log_capacity")
        conf = conf_log[:n].cpu().numpy()
        cls = cls_log[:n].cpu().numpy()
        for c in range(self.nc):
            s = np.sort(conf[cls == c])[::-1]
            lists[c] = [float(v) for v in s]
        return lists

    @staticmethod
    def gmm_policy(scores, given_gt_thr=0.5, policy='high'):
        """:134-186: threshold = lowest score of the positive mixture component that is at least as high as the component's
        most likely sample"""
        if len(scores) < 4:
            return given_gt_thr
        import sklearn.mixture as skm
        scores = np.asarray(scores)
        if scores.ndim == 1:
            scores = scores[:, np.newaxis]
        gmm = skm.GaussianMixture(2, weights_init=[1 / 2, 1 / 2], means_init=[[np.min(scores)], [np.max(scores)]],
                                  precisions_init=[[[1.0]], [[1.0]]])
        gmm.fit(scores)
        assign = gmm.predict(scores)
        logp = gmm.score_samples(scores)
        assert policy in ('middle', 'high')
        if not (assign == 1).any():
            return given_gt_thr
        if policy == 'high':
            logp[assign == 0] = -np.inf
            top = np.argmax(logp, axis=0)
            pos = (assign == 1) & (scores >= scores[top]).squeeze()
            thr = float(scores[pos].min())
        else:
            thr = float(scores[assign == 1].min())
        return max(given_gt_thr, thr)

    def update_epoch_cls_thr(self, epoch):
        """:188-240"""
        lists = self.epoch_scores()
        if self._cls_hist is not None:
            self.cls_tmp = self._cls_hist.cpu().numpy().astype(np.float64)
        for c in range(self.nc):
            s = lists[c]
            self.cls_num_total[c] += len(s)
            max_pseudo_label_num = int(self.cls_num_total[c] / (epoch + 1))
            if len(s) == 0:
                self.cls_thr_high[c] = self.ignore_thres_high
                self.cls_thr_low[c] = self.ignore_thres_low
                pos_loc_high = pos_loc_low = -1
            else:
                pos_loc_high = int(len(s) * self.resample_high_percent)
                pos_loc_low = min(max_pseudo_label_num, int(len(s) * self.resample_low_percent))
                self.cls_thr_high[c] = self.gmm_policy(np.array(s), given_gt_thr=0.0, policy='high')
                self.cls_thr_low[c] = max(self.ignore_thres_low, s[pos_loc_low])
            LOGGER.info(f'{len(s)} {max_pseudo_label_num} {pos_loc_high}:{self.cls_thr_high[c]} {pos_loc_low}:{self.cls_thr_low[c]}')
        if self._log is not None:
            self._log[2].zero_()
        self._cls_hist = None
        self.cls_tmp = np.zeros(self.nc)
        self.count = 0
        self.pse_count = 0
