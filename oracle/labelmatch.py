"""TEST INFRASTRUCTURE -- CPU restatement of the reference's ``LabelMatch`` (utils/labelmatch.py:56-354).

  * ``create_pseudo_label_online_with_gt`` (:271-354): NMS, every detection's confidence appended to the per-class
    ``score_list_epoch`` (:279-287), the FairPseudoLabel warp/filter plus a clip of the normalised xywh (:333);
  * ``gmm_policy`` (:134-186) and ``update_epoch_cls_thr`` (:188-240): per-class thresholds at the end of an epoch.

``sklearn.mixture.GaussianMixture`` is the reference's own dependency for the mixture fit (weights / means / precisions
are all initialised explicitly, so the fit is deterministic); it is used here as it is there.
"""
import numpy as np

from . import nms as _nms
from . import pseudo_label as _pl


class LabelMatchState:
    def __init__(self, nc, ignore_thres_low, ignore_thres_high, resample_high_percent=0.0, resample_low_percent=0.0):
        self.nc = nc
        self.ignore_thres_low, self.ignore_thres_high = ignore_thres_low, ignore_thres_high
        self.resample_high_percent, self.resample_low_percent = resample_high_percent, resample_low_percent
        self.cls_thr_high = [ignore_thres_high] * nc
        self.cls_thr_low = [ignore_thres_low] * nc
        self.score_list_epoch = [[] for _ in range(nc)]
        self.cls_num_total = np.zeros(nc)

    def create_pseudo_label(self, pred, M_s, width, height, conf_thres, iou_thres):
        dets, _ = _nms.non_max_suppression_ssod(pred, conf_thres, iou_thres)
        for o in dets:
            for row in np.asarray(o, np.float32).reshape(-1, 8):
                self.score_list_epoch[int(row[5])].append(float(row[4]))          # :287
        return _pl.create_pseudo_label(dets, M_s, width, height, clip01=True)

    def update_epoch_cls_thr(self, epoch):
        for c in range(self.nc):
            s = sorted(self.score_list_epoch[c], reverse=True)                     # :210
            self.cls_num_total[c] += len(s)
            max_n = int(self.cls_num_total[c] / (epoch + 1))
            if not s:
                self.cls_thr_high[c] = self.ignore_thres_high
                self.cls_thr_low[c] = self.ignore_thres_low
            else:
                pos_low = min(max_n, int(len(s) * self.resample_low_percent))      # :223
                self.cls_thr_high[c] = gmm_policy(np.array(s), given_gt_thr=0.0, policy="high")
                self.cls_thr_low[c] = max(self.ignore_thres_low, s[pos_low])
        self.score_list_epoch = [[] for _ in range(self.nc)]


def gmm_policy(scores, given_gt_thr=0.5, policy="high"):
    if len(scores) < 4:
        return given_gt_thr
    import sklearn.mixture as skm
    scores = np.asarray(scores)
    if scores.ndim == 1:
        scores = scores[:, None]
    gmm = skm.GaussianMixture(2, weights_init=[0.5, 0.5], means_init=[[scores.min()], [scores.max()]],
                              precisions_init=[[[1.0]], [[1.0]]])
    gmm.fit(scores)
    assign = gmm.predict(scores)
    ll = gmm.score_samples(scores)
    if not (assign == 1).any():
        return given_gt_thr
    if policy == "high":
        ll[assign == 0] = -np.inf
        k = int(np.argmax(ll))
        pos = (assign == 1) & (scores >= scores[k]).squeeze()
        thr = float(scores[pos].min())
    else:
        thr = float(scores[assign == 1].min())
    return max(given_gt_thr, thr)
