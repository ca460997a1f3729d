"""Host-side mirror of the reference's ``utils/general.py`` for the SSOD hot path.

Same names / argument meaning / return types as the reference functions the trainers call
(SURVEY.md section 8b); the arithmetic runs in the gfx950 kernels behind include/et_hip.h.
"""
import ctypes

import torch

from .. import _lib

MAX_WH = 7680  # utils/general.py:907, :1013
MAX_NMS = 30000  # utils/general.py:908, :1014


def nms_ssod_padded(prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False, max_det=300):
    """Device-resident form of ``non_max_suppression_ssod`` (no host synchronisation).

    prediction (B, A, 5+nc) fp32 ->
      dets (B, max_det, 8) [x1,y1,x2,y2,conf,cls,obj_conf,cls_conf] zero padded,
      counts (B,) int32, keep (B, max_det) int64 (index into the reference's pre-NMS candidate
      matrix, -1 padded), n_candidates (B,) int32.
    """
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if prediction.dtype != torch.float32:
        prediction = prediction.float()
    prediction = prediction.contiguous()
    B, A, no = prediction.shape
    dev = prediction.device
    lib = _lib.load()
    nbytes = ctypes.c_size_t()
    _lib.check(lib.et_nms_ssod_workspace_bytes(B, A, ctypes.byref(nbytes)), "et_nms_ssod_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    dets = torch.empty((B, max_det, 8), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    keep = torch.empty((B, max_det), dtype=torch.int64, device=dev)
    ncand = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(lib.et_nms_ssod(_lib.ptr(prediction), B, A, no, conf_thres, iou_thres, int(bool(agnostic)),
                               max_det, _lib.ptr(dets), _lib.ptr(counts), _lib.ptr(keep), _lib.ptr(ncand),
                               _lib.ptr(ws), nbytes.value, _lib.stream(prediction)), "et_nms_ssod")
    return dets, counts, keep, ncand


def _with_apriori_labels(prediction, labels):
    """utils/general.py:924-931 (and :1027-1034): per-image (n, 5) [cls, x, y, w, h] boxes join the candidates with obj = 1 and a
    one-hot class row, BEHIND the image's own anchors (the order the reference's torch.cat gives, which the stable sort of the NMS
    sees).  Images with fewer labels are padded with all-zero rows: obj = 0 never passes `obj > conf_thres`."""
    B, A, no = prediction.shape
    nc = no - 5
    n_max = max((len(l) for l in labels), default=0)
    if n_max == 0:
        return prediction
    extra = torch.zeros((B, n_max, no), dtype=prediction.dtype, device=prediction.device)
    for i, l in enumerate(labels):
        if len(l):
            l = torch.as_tensor(l, dtype=prediction.dtype, device=prediction.device)
            extra[i, :len(l), :4] = l[:, 1:5]
            extra[i, :len(l), 4] = 1.0
            extra[i, torch.arange(len(l), device=prediction.device), l[:, 0].long() + 5] = 1.0
    return torch.cat((prediction, extra), 1)


def non_max_suppression_ssod(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                             num_points=0, multi_label=False, labels=(), max_det=300):
    """Runs Non-Maximum Suppression (NMS) on inference results (reference utils/general.py:887).

    Returns:
         list of detections, on (n,8) tensor per image [xyxy, conf, cls, obj_conf, cls_conf]
         ((n,6) [xyxy, conf, cls] with multi_label, as the reference's :948-950 builds it)

    The SSOD configs call this with the defaults (one kernel family, et_nms_ssod).  The optional arguments are served by the same
    kernels: `labels` appends rows to the prediction; `classes` (single label) clears the objectness of the boxes whose best
    class is not listed, which removes exactly the rows :958-959 drops; `multi_label` is the val path's kernel (et_nms) -- its extra
    candidate test `max cls > conf_thres` (:1002) never removes a row that :948 would emit while obj <= 1 (sigmoid outputs).
    """
    if num_points:
        raise NotImplementedError("key points (num_points > 0) are outside the SSOD hot path")
    if prediction.dtype != torch.float32:
        prediction = prediction.float()
    nc = prediction.shape[2] - 5
    if labels and any(len(l) for l in labels):
        prediction = _with_apriori_labels(prediction, labels)
    if multi_label and nc > 1:
        dets, counts, _, _ = nms_padded(prediction, conf_thres, iou_thres, classes, agnostic, True, max_det)
        counts = counts.tolist()
        return [dets[i, :n] for i, n in enumerate(counts)]
    if classes is not None:
        # best class of a box as :952 finds it (first maximum of cls * obj); a box of another class leaves the candidate set
        best = (prediction[..., 5:5 + nc] * prediction[..., 4:5]).argmax(-1)
        allowed = torch.isin(best, torch.as_tensor(list(classes), device=prediction.device, dtype=best.dtype))
        prediction = prediction.clone()
        prediction[..., 4] = torch.where(allowed, prediction[..., 4], torch.zeros_like(prediction[..., 4]))
    dets, counts, _, _ = nms_ssod_padded(prediction, conf_thres, iou_thres, agnostic, max_det)
    counts = counts.tolist()  # the one host sync of the list-returning API
    return [dets[i, :n] for i, n in enumerate(counts)]


def nms_padded(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
               max_det=300, max_nms=MAX_NMS):
    """Device-resident form of ``non_max_suppression`` (no host synchronisation): dets (B, max_det, 6)
    [x1,y1,x2,y2,conf,cls] zero padded, counts (B,) int32, keep (B, max_det) int64, n_candidates (B,) int32."""
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if prediction.dtype != torch.float32:
        prediction = prediction.float()
    prediction = prediction.contiguous()
    B, A, no = prediction.shape
    dev = prediction.device
    mask = (1 << 128) - 1
    if classes is not None:                                 # utils/general.py:1061
        mask = 0
        for c in classes:
            if 0 <= int(c) < 128:
                mask |= 1 << int(c)
    lib = _lib.load()
    nbytes = ctypes.c_size_t()
    ml = int(bool(multi_label))
    _lib.check(lib.et_nms_workspace_bytes(B, A, no, ml, max_nms, ctypes.byref(nbytes)), "et_nms_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    dets = torch.empty((B, max_det, 8), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    keep = torch.empty((B, max_det), dtype=torch.int64, device=dev)
    ncand = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(lib.et_nms(_lib.ptr(prediction), B, A, no, conf_thres, iou_thres, int(bool(agnostic)), ml,
                          mask & 0xFFFFFFFFFFFFFFFF, mask >> 64, max_nms, float(MAX_WH), max_det, _lib.ptr(dets),
                          _lib.ptr(counts), _lib.ptr(keep), _lib.ptr(ncand), _lib.ptr(ws), nbytes.value,
                          _lib.stream(prediction)), "et_nms")
    return dets[..., :6], counts, keep, ncand


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300):
    """Runs Non-Maximum Suppression (NMS) on inference results (reference utils/general.py:994; val.py:335 calls
    it with multi_label=True).

    Returns:
         list of detections, on (n,6) tensor per image [xyxy, conf, cls]
    """
    if labels and any(len(l) for l in labels):          # autolabelling, utils/general.py:1027-1034
        prediction = _with_apriori_labels(prediction.float(), labels)
    dets, counts, _, _ = nms_padded(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det)
    counts = counts.tolist()  # the one host sync of the list-returning API
    return [dets[i, :n] for i, n in enumerate(counts)]


def xywh2xyxy(x):
    # Convert nx4 boxes from [x, y, w, h] to [x1, y1, x2, y2] (reference utils/general.py:630)
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def xyxy2xywh(x):
    # Convert nx4 boxes from [x1, y1, x2, y2] to [x, y, w, h] (reference utils/general.py:549)
    y = x.clone()
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def make_divisible(x, divisor):
    # Returns x evenly divisible by divisor (reference utils/general.py:470)
    import math
    return math.ceil(x / divisor) * divisor
