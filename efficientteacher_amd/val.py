"""Per-batch inference path of the reference's validation loop (reference val.py:277-338):

    img uint8 -> (half | float) / 255 -> model(img) in eval mode (BatchNorm folded into the conv epilogues)
              -> non_max_suppression(out, conf_thres=0.001, iou_thres=0.6, multi_label=True, agnostic=single_cls)

``infer_batch`` is that sequence on the MI355X kernels: the uint8 batch is normalised inside the input pack kernel, the
EMA / student detector runs its folded-BN inference convs (bf16 when ``half``: the reference's fp16 switch maps to this
package's bf16 compute mode), the decoded (B, A, 5+nc) tensor goes through et_nms (multi-label candidates, exact
max_nms cut, class-offset NMS) and comes back as the reference's ``list[Tensor(n, 6)]``.  The mAP bookkeeping around it
(val.py:339-420: ConfusionMatrix, ap_per_class, COCO json) is host code of the reference and stays there.
"""
import torch

from .utils.general import non_max_suppression


@torch.no_grad()
def infer_batch(model, img, conf_thres=0.001, iou_thres=0.6, half=True, augment=False, single_cls=False, multi_label=True,
                max_det=300, labels=()):
    """One batch of val.run: returns (detections list[Tensor(n,6)] [x1,y1,x2,y2,conf,cls], train_out) -- train_out are the raw
    head outputs the reference feeds to compute_loss (val.py:312-314)."""
    if labels:
        raise NotImplementedError("save_hybrid autolabelling (labels=lb) stays on the reference's host path")
    was_training = model.training
    dtype = torch.bfloat16 if half else torch.float32
    inner = model.module if hasattr(model, "module") else model
    if inner._compute_dtype != dtype:
        inner.set_compute_dtype(dtype)
    model.eval()
    if img.dtype != torch.uint8:                      # already normalised by the caller (val.py:283-289 did img /= 255)
        img = img.float()
    outputs = model(img, augment=augment)
    # val.py:300-318 "ugly solution": SSOD detectors return ((z, train_out), feats), plain ones (z, train_out)
    out = outputs
    train_out = None
    while isinstance(out, (tuple, list)):
        if len(out) == 2 and torch.is_tensor(out[0]) and out[0].dim() == 3:
            out, train_out = out[0], out[1]
            break
        out = out[0]
    dets = non_max_suppression(out, conf_thres, iou_thres, multi_label=multi_label, agnostic=single_cls, max_det=max_det)
    if was_training:
        model.train()
    return dets, train_out
