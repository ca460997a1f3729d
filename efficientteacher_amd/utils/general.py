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


def non_max_suppression_ssod(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                             num_points=0, multi_label=False, labels=(), max_det=300):
    """Runs Non-Maximum Suppression (NMS) on inference results (reference utils/general.py:887).

    Returns:
         list of detections, on (n,8) tensor per image [xyxy, conf, cls, obj_conf, cls_conf]
    """
    if classes is not None or num_points or labels:
        raise NotImplementedError("classes / num_points / labels are outside the SSOD hot path")
    nc = prediction.shape[2] - 5
    if multi_label and nc > 1:
        raise NotImplementedError("the reference's non_max_suppression_ssod never runs multi_label on the SSOD path "
                                  "(configs: multi_label False); use non_max_suppression for the val.py path")
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
    if labels:
        raise NotImplementedError("apriori `labels` (autolabelling, utils/general.py:1027-1034) is outside the path")
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
