"""TEST INFRASTRUCTURE -- CPU restatement of the online pseudo-label creator.

Restates ``FairPseudoLabel.create_pseudo_label_online_with_gt``
(utils/self_supervised_utils.py:194-245) and its helpers
``output_to_target_ssod`` (utils/plots.py:485-491), ``online_label_transform``
(utils/self_supervised_utils.py:414-454), ``box_candidates`` (:316-321), in numpy.

Precision contract of the reference (reproduced exactly):
  * NMS rows are fp32; ``xyxy2xywh`` at plots.py:490 runs on the fp32 row, the
    result is then widened to fp64 (np.array of a mixed python-int / np.float32 list);
  * everything after that (xywh2xyxy, affine warp, clip, candidate filter, xyxy2xywh,
    normalise, flips) is fp64.
"""
import numpy as np

from . import nms as _nms


def output_to_target_ssod(dets):
    """list of (n_i,8) fp32 [xyxy,conf,cls,obj,clsconf] -> (N,9) fp64
    [img, cls, x, y, w, h, conf, obj, clsconf] (pixels)."""
    rows = []
    for i, o in enumerate(dets):
        o = np.asarray(o, np.float32).reshape(-1, 8)
        if not o.shape[0]:
            continue
        xywh = _nms.xyxy2xywh(o[:, :4])                   # fp32 arithmetic
        r = np.empty((o.shape[0], 9), np.float64)
        r[:, 0] = i
        r[:, 1] = o[:, 5]
        r[:, 2:6] = xywh
        r[:, 6] = o[:, 4]
        r[:, 7] = o[:, 6]
        r[:, 8] = o[:, 7]
        rows.append(r)
    return np.concatenate(rows, 0) if rows else np.zeros((0,), np.float64)


def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.1, eps=1e-16):
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def transform_image_targets(t, M, s, width, height):
    """t (n,8) fp64 [cls, x1,y1,x2,y2, conf,obj,clsconf] -> filtered, warped (n',8)."""
    n = len(t)
    if not n:
        return t
    xy = np.ones((n * 4, 3))
    xy[:, :2] = t[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
    xy = xy @ M.T
    xy = xy[:, :2].reshape(n, 8)
    x = xy[:, [0, 2, 4, 6]]
    y = xy[:, [1, 3, 5, 7]]
    new = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
    new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
    new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
    i = box_candidates(box1=t[:, 1:5].T * s, box2=new.T, area_thr=0.10)
    t = t[i]
    t[:, 1:5] = new[i]
    return t


def create_pseudo_label(dets, M_s, width, height, clip01=False):
    """dets: NMS output list; M_s (B,13) fp64 [img, M00..M22, s, ud, lr].
    Returns (targets (N',9) fp64 normalised, invalid_target_shape).
    clip01: LabelMatch's variant (utils/labelmatch.py:333) clips the normalised xywh to [0, 1] before the flips."""
    tnp = output_to_target_ssod(dets)
    out = []
    M_s = np.asarray(M_s, np.float64)
    if tnp.ndim == 2 and tnp.shape[0] > 0:
        for i in range(len(dets)):
            it = tnp[tnp[:, 0] == i].copy()
            it[:, 2:6] = _nms.xywh2xyxy(it[:, 2:6])
            row = M_s[M_s[:, 0] == i][0]
            M = row[1:10].reshape(3, 3)
            s, ud, lr = float(row[10]), int(row[11]), int(row[12])
            it = transform_image_targets(it[:, 1:].copy(), M, s, width, height)
            if it.shape[0]:
                it = np.concatenate((np.ones((it.shape[0], 1)) * i, it), 1)
                it[:, 2:6] = _nms.xyxy2xywh(it[:, 2:6])
                it[:, [3, 5]] /= height
                it[:, [2, 4]] /= width
                if clip01:
                    it[:, 2:6] = it[:, 2:6].clip(0, 1)
                if ud == 1:
                    it[:, 3] = 1 - it[:, 3]
                if lr == 1:
                    it[:, 2] = 1 - it[:, 2]
                out.append(it)
    if out:
        return np.concatenate(out, 0), False
    return np.zeros((0, 9), np.float64), True
