"""Host-side mirror of the reference's ``models/loss/loss.py`` for the SSOD hot path.

``ComputeLoss(model, cfg)(p, targets) -> (loss*bs [1], dict(box, obj, cls, loss))`` with the
reference's constructor logic (models/loss/loss.py:95-136); the arithmetic (anchor assignment,
CIoU, BCE, scatter, reductions AND their gradients) runs in ``et_yolo_loss`` (csrc/loss.hip).
"""
import torch

from ... import ops
from ...utils.torch_utils import is_parallel


def smooth_BCE(eps=0.1):  # reference models/loss/loss.py:26
    return 1.0 - 0.5 * eps, 0.5 * eps


class YoloLossFn(torch.autograd.Function):
    """out (8,) = [lbox*w, lobj*w, lcls*w, loss*bs, n_pos pass0..3]; d out[3] / d p from the fused kernel."""

    @staticmethod
    def forward(ctx, table, hp, anchors_host, balance, *p):
        hp = dict(hp)
        ctx.grad_dst = hp.pop("grad_dst", None) or [None] * len(p)
        ota = hp.pop("ota", None)
        if ota is None:
            out, dps = ops.yolo_loss(list(p), table, anchors_host, balance, **hp)
        else:
            # ComputeLoss.ota_loss (loss.py:210-303): SimOTA-matched positives with the objectness read from the last
            # channel (:246), then the plain anchor-based half (:251-292); the three sums are added BEFORE the weights,
            # which is what adding the two weighted results amounts to.  Both halves accumulate into the same gradient.
            pl = list(p)
            match = ops.ota_assign(pl, table, anchors_host, ota["strides"], nc=hp["nc"], anchor_t=hp["anchor_t"],
                                   top_k=ota["top_k"], img_size=ota["img_size"])
            o1, dps = ops.yolo_loss(pl, table, anchors_host, balance, ota_match=match, obj_channel=hp["nc"] + 4, **hp)
            o2, dps = ops.yolo_loss(pl, table, anchors_host, balance, dps=dps, **hp)
            out = o1 + o2
        ctx.dps = dps
        ctx.meta = [(pi.shape, pi.stride(), pi.dtype) for pi in p]
        ctx.bs = p[0].shape[0]
        ctx.mark_non_differentiable()
        return out

    @staticmethod
    def backward(ctx, gout):
        g3 = gout[3:4].contiguous().float()
        grads = []
        for dp, (shape, stride, dtype), dst in zip(ctx.dps, ctx.meta, ctx.grad_dst):
            out = None
            if dst is not None:          # this tensor is one batch half of a split head output: write in place
                holder, half, n = dst
                flat = holder.flat(half, n)
                if flat.numel() == dp.numel() and holder.dtype == dtype:
                    out = flat
            g = ops.scale_cast(dp, dtype, scale=float(ctx.bs), dev_scale=g3, out=out)
            grads.append(g.as_strided(shape, stride))
        return (None, None, None, None, *grads)


def _head_of(model):
    return model.module.head if is_parallel(model) else model.head


class ComputeLoss:
    # Compute losses (reference models/loss/loss.py:93)
    def __init__(self, model, cfg):
        self.sort_obj_iou = False
        self.fl_gamma = float(cfg.Loss.fl_gamma)      # > 0: FocalLoss around BCEcls / BCEobj (reference loss.py:112-114)
        self.cls_pw, self.obj_pw = float(cfg.Loss.cls_pw), float(cfg.Loss.obj_pw)
        self.cp, self.cn = smooth_BCE(eps=cfg.Loss.label_smoothing)
        det = _head_of(model)
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        # Loss.autobalance (reference loss.py:118, :193-197): the balance weights follow 1 / (objectness loss of the level) and
        # are renormalised by the stride-16 level after every call -- kept in device memory and updated by the loss kernel
        # itself (the reference reads obji.item() per level on the host)
        self.autobalance = bool(cfg.Loss.autobalance)
        self.ssi = [float(s) for s in det.stride].index(16.0) if self.autobalance else 0
        self._balance_dev = None
        self.gr = 1.0
        nl = det.nl
        nc = 1 if cfg.single_cls else cfg.Dataset.nc
        self.box_w = cfg.Loss.box * 3.0 / nl
        self.obj_w = cfg.Loss.obj
        self.cls_w = cfg.Loss.cls * nc / 80. * 3. / nl
        self.anchor_t = cfg.Loss.anchor_t
        self.single_targets = cfg.Loss.single_targets
        for k in 'na', 'nc', 'nl', 'num_keypoints', 'anchors':
            setattr(self, k, getattr(det, k))
        if self.num_keypoints > 0:
            raise NotImplementedError("keypoint losses are outside the hot path")
        # Loss.assigner_type == 'SimOTA' (loss.py:131-136, :306): dynamic-k matched positives on top of the anchor-based ones
        self.ota = cfg.Loss.assigner_type == 'SimOTA'
        self.top_k = int(cfg.Loss.top_k)
        self._strides = [float(s) for s in det.stride]
        self._anchors_host = [[[float(v) for v in a] for a in lvl] for lvl in det.anchors.detach().cpu().tolist()]

    def _hp(self):
        return dict(nc=self.nc, anchor_t=float(self.anchor_t), gr=float(self.gr), cp=float(self.cp), cn=float(self.cn),
                    cls_pw=self.cls_pw, obj_pw=self.obj_pw, box_w=float(self.box_w), obj_w=float(self.obj_w),
                    cls_w=float(self.cls_w), fl_gamma=self.fl_gamma)

    def default_loss(self, p, targets, table=None, ota=False):
        """table: a ready (NT, 8) device table [img, cls, x, y, w, h, score, flags] (rows with flags 0 are padding) instead
        of `targets` -- the fixed-capacity form the captured step graph replays (trainer/graph_step.py)."""
        dev = p[0].device
        if table is None:
            t = targets[:, :6].to(device=dev, dtype=torch.float32)
            n = t.shape[0]
            table = torch.cat((t, torch.zeros((n, 1), device=dev), torch.ones((n, 1), device=dev)), 1)
        hp = self._hp()
        hp["grad_dst"] = [getattr(pi, "_et_grad_dst", None) for pi in p]
        if ota:
            if self.top_k > 13:
                raise NotImplementedError("Loss.top_k > 13: the matching kernel keeps 13 candidates per thread")
            # yolo_anchor_assigner.py:128 scales the targets by a literal 640 ("TODO" there); kept as is
            hp["ota"] = dict(strides=self._strides, top_k=self.top_k, img_size=640.0)
        if self.autobalance:
            if ota:
                raise NotImplementedError("Loss.autobalance with SimOTA: the reference updates the weights inside the first half only")
            if self._balance_dev is None or self._balance_dev.device != dev:
                self._balance_dev = torch.tensor(self.balance, dtype=torch.float32, device=dev)
            hp["balance_dev"], hp["ssi"] = self._balance_dev, self.ssi
        out = YoloLossFn.apply(table, hp, self._anchors_host, self.balance, *p)
        loss = out[3:4]
        det = out.detach()
        loss_dict = dict(box=det[0:1], obj=det[1:2], cls=det[2:3], loss=det[3:4])
        return loss, loss_dict

    def ota_loss(self, p, targets, table=None):
        return self.default_loss(p, targets, table=table, ota=True)

    def __call__(self, p, targets):
        return self.ota_loss(p, targets) if self.ota else self.default_loss(p, targets)


class _DomainFocal:
    """Shared body of DomainLoss / TargetLoss (reference models/loss/loss.py:376-421): the three netD
    outputs (B,2,H,W) -> 0.5 * softmax-focal loss against a constant domain label."""
    label = 0

    def __call__(self, feature):
        from ...autograd import DomainFocalFn
        feats = [f.permute(0, 2, 3, 1) for f in feature]      # NHWC views of the netD results
        return DomainFocalFn.apply(self.label, *feats)[0]


class DomainLoss(_DomainFocal):     # source domain: label 0 (loss.py:398)
    label = 0


class TargetLoss(_DomainFocal):     # target domain: label 1 (loss.py:376)
    label = 1
