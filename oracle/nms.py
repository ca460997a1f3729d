"""TEST INFRASTRUCTURE -- CPU restatement of the pseudo-label filter.

Restates, in numpy fp32:
  * ``torchvision.ops.nms``  (third party, requirements.txt:12 torchvision>=0.8.1;
    call site utils/general.py:976).  PARITY UNPINNED: torchvision is not
    installed and the reference holds no golden output for it.  Published
    kernel semantics (torchvision/csrc/ops/cpu/nms_kernel.cpp): boxes sorted by
    score, *stable*, descending; box j is dropped iff a kept higher-ranked box i
    has inter/(area_i+area_j-inter) > thr (strict), area=(x2-x1)*(y2-y1),
    inter=max(0,xx2-xx1)*max(0,yy2-yy1), all fp32; returns int64 indices into
    the input in descending-score order.
    WHICH THRESHOLD COMPARE IS PINNED: the fp32 one, ``ovr > float(thr)`` -- the
    rule of torchvision's *CUDA* kernel (torchvision/csrc/ops/cuda/nms_kernel.cu:
    ``devIoU(a, b, const float threshold)``), the kernel the reference trains on
    (utils/general.py:976 runs on the model's device).  The CPU kernel keeps
    ``iou_threshold`` a double and compares ``double(ovr) > thr``.  The two differ
    ONLY when ovr == fp32(thr) exactly and fp32(thr) > thr: 0.6 (val.py's
    setting) is such a value (fp32(0.6) = 0.60000002...: the CPU rule suppresses
    an exact tie, the CUDA rule keeps it), 0.65 (SSOD.nms_iou_thres) and 0.45 are
    not.  ``nms(..., thr_compare="cpu_double")`` restates the other rule; the
    golden case ``tie06`` (tests/golden/nms.npz) holds an exact tie at 0.6 and the
    keep set under that rule.  ``tie06`` is ORACLE-DERIVED, not a reference output:
    torchvision is installed neither in the build container nor on the GPU box, so
    when oracle/make_golden.py runs the reference, ``torchvision.ops.nms`` is THIS
    restatement (oracle/ref_loader.py) and its ``ref == mine`` check compares the
    restatement with itself.  The rule above is a reading of torchvision's source,
    UNVERIFIED against a real torchvision; tests/test_nms_torchvision.py is the pin
    and reports SKIPPED wherever torchvision is absent (parity
    unpinned at this boundary).
  * ``non_max_suppression_ssod``  utils/general.py:887-992
  * ``non_max_suppression``       utils/general.py:994-1100  (val.py path, row f-1)
  * ``xywh2xyxy`` / ``xyxy2xywh`` utils/general.py:630-637 / 549-556
"""
import numpy as np

MAX_WH = 7680.0  # utils/general.py:907
MAX_NMS = 30000  # utils/general.py:908
F32 = np.float32


def xywh2xyxy(x):
    y = np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def xyxy2xywh(x):
    y = np.copy(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def nms(boxes, scores, iou_thres, thr_compare="cuda_fp32"):
    """Greedy NMS, fp32, returns int64 keep indices (descending score).  thr_compare: "cuda_fp32" (pinned: threshold rounded to
    fp32, torchvision's CUDA kernel) or "cpu_double" (torchvision's CPU kernel: fp32 overlap promoted to double) -- see the header."""
    assert thr_compare in ("cuda_fp32", "cpu_double")
    boxes = np.ascontiguousarray(boxes, dtype=F32)
    scores = np.ascontiguousarray(scores, dtype=F32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[order, k] for k in range(4))
    areas = (x2 - x1) * (y2 - y1)
    thr = F32(iou_thres) if thr_compare == "cuda_fp32" else np.float64(iou_thres)
    suppressed = np.zeros(n, bool)
    keep = []
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(n):
            if suppressed[i]:
                continue
            keep.append(order[i])
            if i + 1 == n:
                break
            xx1 = np.maximum(x1[i], x1[i + 1:])
            yy1 = np.maximum(y1[i], y1[i + 1:])
            xx2 = np.minimum(x2[i], x2[i + 1:])
            yy2 = np.minimum(y2[i], y2[i + 1:])
            w = np.maximum(F32(0), xx2 - xx1)
            h = np.maximum(F32(0), yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
            suppressed[i + 1:] |= (ovr > thr) if thr_compare == "cuda_fp32" else (ovr.astype(np.float64) > thr)
    return np.asarray(keep, np.int64)


def nms_torch(boxes, scores, iou_threshold):
    """torch-tensor wrapper used as the ``torchvision.ops.nms`` stub."""
    import torch

    k = nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(iou_threshold))
    return torch.from_numpy(k).to(boxes.device)


def non_max_suppression_ssod(prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False,
                             max_det=300, classes=None, multi_label=False, labels=()):
    """utils/general.py:887-992 (num_points = 0).

    prediction (B, A, 5+nc) fp32 -> list of B arrays (n_i, 8)
    [x1,y1,x2,y2, conf, cls, obj_conf, cls_conf] -- (n_i, 6) [x1,y1,x2,y2, conf, cls] with multi_label (:948-950);
    also returns the keep indices (into the pre-NMS candidate rows) so tests can check indices bit-exactly.
    classes: keep only these class ids (:958-959); labels: per-image (n, 5) [cls, x, y, w, h] apriori boxes appended to
    the candidates with obj = cls = 1 (:924-931).
    """
    prediction = np.asarray(prediction, dtype=F32)
    nc = prediction.shape[2] - 5
    ct = F32(conf_thres)
    multi_label = multi_label and nc > 1                  # :912
    width = 6 if multi_label else 8
    out, keeps = [], []
    for xi, x in enumerate(prediction):
        x = x[x[:, 4] > ct].copy()                       # :921 obj filter
        if labels and len(labels[xi]):                    # :924-931
            l = np.asarray(labels[xi], dtype=F32)
            v = np.zeros((len(l), nc + 5), F32)
            v[:, :4] = l[:, 1:5]
            v[:, 4] = 1.0
            v[np.arange(len(l)), l[:, 0].astype(np.int64) + 5] = 1.0
            x = np.concatenate((x, v), 0)
        if not x.shape[0]:
            out.append(np.zeros((0, width), F32)); keeps.append(np.zeros((0,), np.int64)); continue
        cls_score = x[:, 5:5 + nc].max(1, keepdims=True)  # :937
        x[:, 5:5 + nc] *= x[:, 4:5]                       # :938
        box = xywh2xyxy(x[:, :4])                         # :943
        if multi_label:                                   # :948-950
            i, j = np.nonzero(x[:, 5:5 + nc] > ct)
            x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(F32)), 1)
        else:
            j = x[:, 5:5 + nc].argmax(1)[:, None]             # :952 (first max index)
            conf = np.take_along_axis(x[:, 5:5 + nc], j, 1)
            obj = x[:, 4:5]
            x = np.concatenate((box, conf, j.astype(F32), obj, cls_score), 1)[conf.reshape(-1) > ct]
        if classes is not None:                           # :958-959
            x = x[np.isin(x[:, 5], np.asarray(classes, F32))]
        n = x.shape[0]
        if not n:
            out.append(np.zeros((0, width), F32)); keeps.append(np.zeros((0,), np.int64)); continue
        if n > MAX_NMS:                                   # :969
            x = x[np.argsort(-x[:, 4], kind="stable")[:MAX_NMS]]
        c = x[:, 5:6] * F32(0 if agnostic else MAX_WH)    # :972
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms(boxes, scores, iou_thres)[:max_det]       # :976-978
        out.append(x[i]); keeps.append(i)
    return out, keeps


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False,
                        multi_label=False, max_det=300, classes=None, max_nms=MAX_NMS):
    """utils/general.py:994-1100 (val.py:335 uses multi_label=True)."""
    prediction = np.asarray(prediction, dtype=F32)
    nc = prediction.shape[2] - 5
    ct = F32(conf_thres)
    multi_label = multi_label and nc > 1
    out = []
    for x in prediction:
        xc = (x[:, 4] > ct) & (x[:, 5:].max(1) > ct)      # :1002
        x = x[xc].copy()
        if not x.shape[0]:
            out.append(np.zeros((0, 6), F32)); continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        if multi_label:
            i, j = np.nonzero(x[:, 5:] > ct)
            x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(F32)), 1)
        else:
            j = x[:, 5:].argmax(1)[:, None]
            conf = np.take_along_axis(x[:, 5:], j, 1)
            x = np.concatenate((box, conf, j.astype(F32)), 1)[conf.reshape(-1) > ct]
        if classes is not None:                            # :1061
            x = x[np.isin(x[:, 5], np.asarray(classes, F32))]
        n = x.shape[0]
        if not n:
            out.append(np.zeros((0, 6), F32)); continue
        if n > max_nms:                                    # :1071 (torch argsort is unstable; pinned as stable)
            x = x[np.argsort(-x[:, 4], kind="stable")[:max_nms]]
        c = x[:, 5:6] * F32(0 if agnostic else MAX_WH)
        i = nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        out.append(x[i])
    return out
