"""Host-side mirror of the reference's ``models/loss/tal_loss.py``: ``ComputeTalLoss(model, cfg)(outputs, targets) ->
(loss [1], dict(loss_iou, loss_dfl, loss_cls, loss, num_fg))`` for the YOLOv8 head.

The arithmetic runs in csrc/tal.hip: the DFL decode of the predicted boxes and the TaskAlignedAssigner (et_tal_assign), then
class / box / DFL terms with their gradients in one pass (et_tal_loss).  The reference file imports two classes that are not
in its tree (gfocal_loss.VarifocalLoss / BboxLoss); what they compute is specified in oracle/v8.py::tal_loss.
"""
import torch

from ... import ops
from ...utils.torch_utils import is_parallel


class _TalLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_scores, pred_distri, aux):
        out, gs, gd = ops.tal_loss(pred_scores, pred_distri, *aux)
        ctx.save_for_backward(gs, gd)
        ctx.dt = (pred_scores.dtype, pred_distri.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        gs, gd = ctx.saved_tensors
        g = gout[3]                                        # only the total is differentiated
        return (gs * g).to(ctx.dt[0]), (gd * g).to(ctx.dt[1]), None


class ComputeTalLoss:
    def __init__(self, model, cfg):
        self.det = model.module.head if is_parallel(model) else model.head
        self.fpn_strides = cfg.Model.Head.strides
        self.grid_cell_size = cfg.Loss.grid_cell_size
        self.grid_cell_offset = cfg.Loss.grid_cell_offset
        self.num_classes = cfg.Dataset.nc
        self.ori_img_size = cfg.Dataset.img_size
        self.use_dfl = cfg.Loss.use_dfl
        self.use_gfl = cfg.Loss.use_gfl
        self.reg_max = cfg.Loss.reg_max
        self.iou_type = cfg.Loss.iou_type
        if not self.use_dfl or self.use_gfl:
            raise NotImplementedError("the fused TAL loss covers use_dfl True / use_gfl False (every shipped YOLOv8 recipe)")
        self.loss_weight = {'class': cfg.Loss.qfl_loss_weight, 'iou': cfg.Loss.box_loss_weight, 'dfl': cfg.Loss.dfl_loss_weight}
        self._anchors = {}

    def _anchor_points(self, shapes, dev):
        key = (tuple(shapes), str(dev))
        a = self._anchors.get(key)
        if a is None:                                       # generate_anchors (nanodet_utils.py:148-180): cell centres in pixels
            pts, st = [], []
            for (h, w), s in zip(shapes, self.fpn_strides):
                sy, sx = torch.meshgrid((torch.arange(h) + self.grid_cell_offset) * s, (torch.arange(w) + self.grid_cell_offset) * s,
                                        indexing="ij")
                pts.append(torch.stack([sx, sy], -1).float().reshape(-1, 2))
                st.append(torch.full((h * w, 1), float(s)))
            a = self._anchors[key] = (torch.cat(pts).to(dev), torch.cat(st).to(dev))
        return a

    def preprocess(self, targets, batch_size):
        """tal_loss.py:131-143: (n,6) [img, cls, x, y, w, h] normalised -> padded (B, G, 5) [cls, x1, y1, x2, y2] pixels"""
        t = targets.detach().to("cpu", torch.float32)
        per = [[] for _ in range(batch_size)]
        for row in t.tolist():
            per[int(row[0])].append(row[1:])
        G = max(max(len(p) for p in per), 0)
        out = torch.zeros(batch_size, G, 5)
        out[..., 0] = -1
        for i, p in enumerate(per):
            if p:
                out[i, :len(p)] = torch.tensor(p)
        box = out[..., 1:5] * float(self.ori_img_size)
        x1, y1 = box[..., 0] - box[..., 2] * 0.5, box[..., 1] - box[..., 3] * 0.5
        out[..., 1:5] = torch.stack([x1, y1, x1 + box[..., 2], y1 + box[..., 3]], -1)
        return out, sum(len(p) for p in per) + batch_size     # the reference counts its dummy row per image

    def assigner_inputs(self, feats, pred_scores, pred_distri):
        """what the TaskAlignedAssigner sees (detached, tal_loss.py:95-102): sigmoid class scores (B,A,nc), predicted xyxy boxes in
        pixels (B,A,4), and the anchor tables -- decoded exactly as the inference head does (DFL expectation, dist2bbox)"""
        dev = pred_scores.device
        B, A, nc = pred_scores.shape
        shapes = [tuple(f.shape[-2:]) for f in feats]
        anchor_points, stride_tensor = self._anchor_points(shapes, dev)
        with torch.no_grad():
            z = torch.empty((B, A, 5 + nc), dtype=torch.float32, device=dev)
            off = 0
            nb = 4 * (self.reg_max + 1)
            for (h, w), s in zip(shapes, self.fpn_strides):
                n = h * w
                ops.v8_decode(pred_distri[:, off:off + n].float().contiguous().view(B, h, w, nb), pred_scores[:, off:off + n].float().contiguous().view(B, h, w, nc),
                              self.reg_max, nc, float(s), self.grid_cell_offset, z, off)
                off += n
            cxcywh = z[..., :4]
            pd_xyxy = torch.cat([cxcywh[..., :2] - cxcywh[..., 2:] / 2, cxcywh[..., :2] + cxcywh[..., 2:] / 2], -1)
        return z[..., 5:], pd_xyxy, anchor_points, anchor_points / stride_tensor, stride_tensor

    def loss_terms(self, pred_scores, pred_distri, anchor_points_s, stride_tensor, tb, ts, fg):
        aux = (anchor_points_s, stride_tensor, tb, ts, fg, self.reg_max, self.iou_type, self.loss_weight['class'],
               self.loss_weight['iou'], self.loss_weight['dfl'])
        return _TalLossFn.apply(pred_scores, pred_distri, aux)

    def __call__(self, outputs, targets):
        feats, pred_scores, pred_distri = outputs
        dev = pred_scores.device
        B, A, nc = pred_scores.shape
        if targets.is_cuda or (dev.type == "cpu" and targets.device.type == "cpu" and ops._lib.is_emulated()):
            # device-resident targets: the padded table is built by a kernel (no .cpu() round trip = no host synchronisation)
            gt_labels, gt_bboxes, mask_gt = ops.tal_targets_pad(targets.to(dev), B, self.ori_img_size, self.ori_img_size)
            num_gts = int(targets.shape[0]) + B              # the reference counts its dummy row per image (:133-137)
        else:
            tg, num_gts = self.preprocess(targets, B)
            tg = tg.to(dev)
            gt_labels, gt_bboxes = tg[..., :1], tg[..., 1:]
            mask_gt = (gt_bboxes.sum(-1, keepdim=True) > 0).float()
        scores, pd_xyxy, anchor_points, anchor_points_s, stride_tensor = self.assigner_inputs(feats, pred_scores, pred_distri)
        with torch.no_grad():
            tl, tb, ts, fg = ops.tal_assign(scores, pd_xyxy, anchor_points, gt_labels, gt_bboxes, mask_gt, topk=13, alpha=1.0, beta=6.0)
        out = self.loss_terms(pred_scores, pred_distri, anchor_points_s, stride_tensor, tb, ts, fg)
        loss = out[3:4]
        d = out.detach()
        return loss, dict(loss_iou=d[0], loss_dfl=d[1], loss_cls=d[2], loss=loss, num_fg=fg.sum() / max(num_gts, 1))
