"""Drop-in under the reference's ``train.py``: the reference's OWN ``Trainer`` / ``SSODTrainer`` with the hot path
replaced (VERDICT r01 "row (b)").

The epoch loop, data loaders, validation, loggers, checkpoint files and callbacks stay the reference's code
(``train()`` trainer/trainer.py:524, ``before_epoch`` :358, ``train_in_epoch`` :406 / ssod_trainer.py:295,
``after_epoch`` :445, ``train_with_unlabeled`` ssod_trainer.py:682): the classes returned by ``hot_path_trainers()``
SUBCLASS the trainers of the tree they are used in and override only

    build_model       this package's Model (same state_dict keys), cfg.weights loaded through utils/checkpoint.py with the
                      reference's intersect / exclude-anchors / strict=False rules (trainer.py:127-144)
    build_optimizer   FlatSGD over the parameter arena, the reference's scheduler and warm-up fields; ``self.scaler`` is a
                      unit scaler (bf16 needs no loss scaling) whose ``scale(loss).backward()`` also finishes the RCCL
                      gradient all-reduce after every backward (DDP semantics, also under gradient accumulation), so the
                      reference's ``update_optimizer`` (trainer.py:381, ssod_trainer.py:458) runs UNCHANGED
    build_ddp_model   FlatDataParallel instead of DistributedDataParallel (trainer.py:313); this package's losses
    train_instance    (SSOD) the device-resident step: teacher on a side stream, padded pseudo labels, no host sync
                      except for the progress-bar numbers the reference prints

train.py edit (reference lines 26-27):

    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    Trainer, SSODTrainer = hot_path_trainers()
"""
import logging

import numpy as np
import torch

from ..parallel import FlatDataParallel
from ..utils.torch_utils import CosineEMA, ModelEMA, SemiSupModelEMA, is_parallel

LOGGER = logging.getLogger(__name__)


class _ReducedLoss:
    """what ``UnitScaler.scale(loss)`` returns: ``backward()`` runs the backward pass and then COMPLETES the data-parallel
    gradient all-reduce -- after EVERY backward, as DistributedDataParallel does (trainer.py:313) and as this package's
    own ``update_optimizer`` does.  With gradient accumulation (accumulate > 1: total batch <= 42, or the warm-up
    interpolation of trainer.py:390) the reference's ``update_optimizer`` reaches ``scaler.step`` only on optimizer-step
    iterations; finishing the collectives there would leave the chunks that the overlap hook launched during an earlier
    micro-step un-waited and marked as launched, so the later micro-steps' conv gradients would never be averaged and a
    collective in flight would race with the next wgrad's writes into the same arena."""

    def __init__(self, loss, trainer):
        self._loss, self._t = loss, trainer

    def backward(self, *a, **k):
        self._loss.backward(*a, **k)
        m = self._t.model
        if isinstance(m, FlatDataParallel):
            m.reduce_gradients()


class UnitScaler:
    """stands in for torch.cuda.amp.GradScaler (trainer.py:248): bf16 activations with fp32 accumulation and fp32 master
    weights need no loss scaling; ``scale(loss).backward()`` is where the data-parallel gradient all-reduce is completed
    (every micro-step, see _ReducedLoss)."""

    def __init__(self, trainer):
        self._t = trainer

    def scale(self, loss):
        return _ReducedLoss(loss, self._t)

    def step(self, optimizer):
        optimizer.step()

    def update(self):
        pass

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class AmpScaler(UnitScaler):
    """the reference's own recipe -- ``amp.GradScaler`` around fp16 autocast (trainer.py:248,348,399-401) -- on this path: fp16 compute
    mode with ``optim.DeviceGradScaler`` (scale / found-inf / growth tracker in device memory, skip and unscale inside the optimizer
    kernels).  ``scale(loss).backward()`` multiplies by the device scale and, as the unit scaler does, completes the data-parallel
    all-reduce, so ``step`` scans the AVERAGED gradient arena: every rank takes the same skip decision without a collective of its own."""

    def __init__(self, trainer, device):
        from ..optim import DeviceGradScaler
        super().__init__(trainer)
        self.dev = DeviceGradScaler(device, enabled=True)

    def scale(self, loss):
        return _ReducedLoss(self.dev.scale(loss), self._t)

    def step(self, optimizer):
        return self.dev.step(optimizer)

    def update(self):
        self.dev.update()

    def get_scale(self):
        return self.dev.get_scale()

    def state_dict(self):
        return self.dev.state_dict()

    def load_state_dict(self, sd):
        self.dev.load_state_dict(sd)


def _intersect(csd, msd, exclude=()):
    # reference utils/torch_utils.py intersect_dicts: same name, same shape, not excluded
    return {k: v for k, v in csd.items() if k in msd and not any(x in k for x in exclude) and tuple(v.shape) == tuple(msd[k].shape)}


class _HotPath:
    """mixin over the reference's Trainer (methods resolved before the reference's through the MRO)"""
    ET_MODEL_MODULE = "efficientteacher_amd.models.detector.yolo"
    # compute dtype on a GPU (hot_path_trainers(compute_dtype=...)): bfloat16 = no loss scaling; float16 = the reference's autocast
    # dtype with the device-resident GradScaler (AmpScaler); float32 = parity mode
    ET_COMPUTE_DTYPE = torch.bfloat16
    # hot_path_trainers(deterministic=True): bit-reproducible BatchNorm statistics in the 16-bit modes (Model.set_deterministic)
    ET_DETERMINISTIC = False

    def _et_model(self, cfg, device):
        import importlib
        model = importlib.import_module(self.ET_MODEL_MODULE).Model(cfg).to(device)
        if self.ET_DETERMINISTIC:
            model._deterministic = True          # picked up by the rebuild inside set_compute_dtype below
        cuda = torch.device(device).type != "cpu"
        model.set_compute_dtype(self.ET_COMPUTE_DTYPE if (cuda or self.ET_COMPUTE_DTYPE == torch.float16) else torch.float32)
        return model

    def _et_load_weights(self, cfg, device):
        from ..utils.checkpoint import load_reference_checkpoint
        ckpt = load_reference_checkpoint(cfg.weights, map_location="cpu")
        exclude = ['anchor'] if not cfg.resume else []
        csd = _intersect(ckpt["model"], self.model.state_dict(), exclude=exclude)
        self.model.load_state_dict(csd, strict=False)
        LOGGER.info(f'Transferred {len(csd)}/{len(self.model.state_dict())} items from {cfg.weights}')
        return ckpt

    def _et_freeze(self, cfg):
        freeze = [f'model.{x}.' for x in range(cfg.freeze_layer_num)]
        for k, v in self.model.named_parameters():
            v.requires_grad = True
            if any(x in k for x in freeze):
                v.requires_grad = False

    def _et_resume(self, cfg, ckpt, emas):
        self.start_epoch = 0
        if ckpt is not None and not getattr(cfg, "reinitial", False):
            for e in emas:
                if e is not None and ckpt.get("ema"):
                    e.ema.load_state_dict(_intersect(ckpt["ema"], e.ema.state_dict()), strict=False)
            if self.ema is not None and ckpt.get("updates") is not None:
                self.ema.updates = ckpt["updates"]
            self.start_epoch = ckpt["epoch"] + 1
            if cfg.resume:
                assert self.start_epoch > 0, f'{cfg.weights} training to {self.epochs} epochs is finished, nothing to resume.'
            if self.epochs < self.start_epoch:
                self.epochs += ckpt["epoch"]
        self.epoch = self.start_epoch
        self.model_type = self.model.model_type
        self.detect = self.model.head

    # ---- trainer.py:125 ---------------------------------------------------------------------------------------
    def build_model(self, cfg, device):
        self.model = self._et_model(cfg, device)
        ckpt = self._et_load_weights(cfg, device) if str(cfg.weights).endswith('.pt') else None
        self._et_freeze(cfg)
        self.ema = ModelEMA(self.model) if self.RANK in [-1, 0] else None
        self._et_resume(cfg, ckpt, [self.ema])
        return ckpt

    # ---- trainer.py:193 ---------------------------------------------------------------------------------------
    def build_optimizer(self, cfg, optinit=True, weight_masks=None, ckpt=None):
        from torch.optim import lr_scheduler
        from ..optim import FlatSGD
        if cfg.Model.RepOpt:
            raise NotImplementedError("RepOptimizer (YOLOv6 re-parameterised training) stays on the reference trainer")
        nbs = 64
        self.accumulate = max(round(nbs / self.batch_size), 1)
        weight_decay = cfg.hyp.weight_decay * self.batch_size * self.accumulate / nbs
        if cfg.adam:
            from ..optim import FlatAdamW
            self.optimizer = FlatAdamW(self.model, lr=cfg.hyp.lr0, betas=(cfg.hyp.momentum, 0.999), weight_decay=weight_decay)
        else:
            self.optimizer = FlatSGD(self.model, lr=cfg.hyp.lr0, momentum=cfg.hyp.momentum, nesterov=True, weight_decay=weight_decay)
        if cfg.linear_lr:
            self.lf = lambda x: (1 - x / (self.epochs - 1)) * (1.0 - cfg.hyp.lrf) + cfg.hyp.lrf
        else:
            import math
            self.lf = lambda x: ((1 - math.cos(x * math.pi / self.epochs)) / 2) * (cfg.hyp.lrf - 1) + 1   # one_cycle(1, lrf, epochs)
        self.scheduler = lr_scheduler.LambdaLR(self.optimizer, lr_lambda=self.lf)
        self.scheduler.last_epoch = self.epoch - 1
        inner = self.model.module if is_parallel(self.model) else self.model
        fp16 = inner.flat_state().compute_dtype == torch.float16
        self.scaler = AmpScaler(self, next(inner.parameters()).device) if fp16 else UnitScaler(self)
        if ckpt is not None and ckpt.get('optimizer') is not None:
            try:
                self.optimizer.load_state_dict(ckpt['optimizer'])
            except (ValueError, KeyError, RuntimeError):
                LOGGER.info('checkpoint optimizer state belongs to another optimizer type: starting it fresh')
        if cfg.SSOD.train_domain and cfg.SSOD.multi_step_lr:      # ssod_trainer.py:86-94
            self.scheduler = lr_scheduler.MultiStepLR(self.optimizer, milestones=cfg.SSOD.milestones, gamma=0.1)
            self.scheduler.last_epoch = self.epoch - 1

    # ---- trainer.py:308 ---------------------------------------------------------------------------------------
    def build_ddp_model(self, cfg, device):
        from ..models.loss import ComputeLoss
        if self.cuda and self.RANK != -1:
            self.model = FlatDataParallel(self.model)
            from .trainer import Trainer as _Core
            _Core._sync_ema_from_rank0(self)          # the EMA copy was taken before the broadcast (see there)
        inner = self.model.module if is_parallel(self.model) else self.model
        inner.nc = self.nc
        inner.names = self.names
        if getattr(self, "dataset", None) is not None and getattr(self.dataset, "labels", None) is not None:
            try:
                from utils.general import labels_to_class_weights       # the reference's own helper (host code)
                inner.class_weights = labels_to_class_weights(self.dataset.labels, self.nc).to(device) * self.nc
            except ImportError:
                pass
        if cfg.Loss.type == 'ComputeTalLoss':          # the YOLOv8 recipes (trainer.py:320-327 dispatches on cfg.Loss.type)
            from ..models.loss import ComputeTalLoss
            self.compute_loss = ComputeTalLoss(self.model, cfg)
        elif cfg.Loss.type == 'ComputeLoss':
            self.compute_loss = ComputeLoss(self.model, cfg)
        else:
            raise NotImplementedError(f"Loss.type {cfg.Loss.type}: the YOLOv5 anchor loss and the YOLOv8 TAL loss are on the "
                                      "MI355X path; other heads keep the reference trainer")
        self.detect = inner.head


class _SsodHotPath(_HotPath):
    ET_MODEL_MODULE = "efficientteacher_amd.models.detector.yolo_ssod"

    # ---- ssod_trainer.py:96 -------------------------------------------------------------------------------------
    def build_model(self, cfg, device):
        self.model = self._et_model(cfg, device)
        ckpt = self._et_load_weights(cfg, device) if str(cfg.weights).endswith('.pt') else None
        self._et_freeze(cfg)
        self.ema = ModelEMA(self.model)
        if cfg.hyp.burn_epochs > 0:
            self.semi_ema = None
        elif self.cosine_ema:
            self.semi_ema = CosineEMA(self.ema.ema, decay_start=cfg.SSOD.ema_rate, total_epoch=self.epochs)
        else:
            self.semi_ema = SemiSupModelEMA(self.ema.ema, cfg.SSOD.ema_rate)
        self._et_resume(cfg, ckpt, [self.ema, self.semi_ema])
        self.extra_teacher_models, self.extra_teacher_class_idxs = [], []
        if len(cfg.SSOD.extra_teachers) > 0:
            raise NotImplementedError("SSOD.extra_teachers: extra teacher ensembles stay on the reference trainer")
        self._side = None
        self.teacher_pred_hook = None
        # the captured-graph step (trainer/graph_step.py) needs this package's update_optimizer; under the reference's own
        # update_optimizer (kept here on purpose) every step is issued eagerly
        self.use_graph, self.graph_warmup, self._eager_steps, self._capturing, self._graph = False, 3, 0, False, None
        return ckpt

    # ---- ssod_trainer.py:258 ------------------------------------------------------------------------------------
    def build_ddp_model(self, cfg, device):
        from ..models.loss import DomainLoss, TargetLoss, build_ssod_loss
        from ..utils.self_supervised_utils import FairPseudoLabel
        super().build_ddp_model(cfg, device)
        self.compute_un_sup_loss = build_ssod_loss(self.model, cfg)
        self.domain_loss = DomainLoss()
        self.target_loss = TargetLoss()
        if cfg.SSOD.pseudo_label_type == 'FairPseudoLabel':
            self.pseudo_label_creator = FairPseudoLabel(cfg)      # the device-resident one (the reference built its own at :68)
        elif cfg.SSOD.pseudo_label_type == 'LabelMatch':          # ssod_trainer.py:70-71; after_epoch (:320-323) drives it unchanged
            from ..utils.labelmatch import LabelMatch
            self.pseudo_label_creator = LabelMatch(cfg, int(len(self.unlabeled_dataset) / self.WORLD_SIZE), self.label_num_per_image,
                                                   cls_ratio_gt=self.cls_ratio_gt)
        else:
            raise NotImplementedError(f"SSOD.pseudo_label_type {cfg.SSOD.pseudo_label_type}")

    # ---- ssod_trainer.py:295 ------------------------------------------------------------------------------------
    def train_in_epoch(self, callbacks):
        """the reference's dispatch between burn-in and SSOD epochs, restated because at the first SSOD epoch it constructs
        the semi-supervised EMA from ITS module's CosineEMA / SemiSupModelEMA (per-tensor deepcopy classes that know nothing
        of the flat arenas and the bf16 shadow); everything it calls is inherited"""
        if self.epoch < self.cfg.hyp.burn_epochs:
            if self.cfg.SSOD.with_da_loss:
                self.train_without_unlabeled_da(callbacks)
            else:
                self.train_without_unlabeled(callbacks)
            if self.RANK in [-1, 0]:
                print('burn_in_epoch: {}, cur_epoch: {}'.format(self.cfg.hyp.burn_epochs, self.epoch))
            return
        if self.epoch == self.cfg.hyp.burn_epochs:             # :305-317 (its state_dict loop there has no effect)
            if self.cosine_ema:
                self.semi_ema = CosineEMA(self.ema.ema, decay_start=self.cfg.SSOD.ema_rate,
                                          total_epoch=self.epochs - self.cfg.hyp.burn_epochs)
            else:
                self.semi_ema = SemiSupModelEMA(self.ema.ema, self.cfg.SSOD.ema_rate)
        self.train_with_unlabeled(callbacks)

    # ---- ssod_trainer.py:568 / :587 -----------------------------------------------------------------------------
    def split_predict_and_feature(self, total_pred, total_feature, n_img):
        from .ssod_trainer import SSODTrainer as _Core
        return _Core.split_predict_and_feature(total_pred, total_feature, n_img)

    def train_instance(self, imgs, targets, paths, unlabeled_imgs, unlabeled_imgs_ori, unlabeled_gt, unlabeled_M, ni, pbar=None,
                       callbacks=None):
        from .ssod_trainer import SSODTrainer as _Core
        items = _Core._train_instance_eager(self, imgs, targets, paths, unlabeled_imgs, unlabeled_imgs_ori, unlabeled_gt, unlabeled_M,
                                            ni, pbar, callbacks)
        if self.RANK in [-1, 0] and getattr(self, "meter", None) is not None:      # ssod_trainer.py:653-678 (the numbers it prints)
            self.meter.update(items)
            self.meter.update(self._pseudo_label_hit_rate(unlabeled_gt))
            if pbar is not None and hasattr(pbar, "set_description"):
                n = len(self.meter.meters.items())
                mem = f'{torch.cuda.memory_reserved() / 1E9 if torch.cuda.is_available() else 0:.3g}G'
                pbar.set_description(('%10s' * 2 + '%10.4g' * (n + 2)) % (f'{self.epoch}/{self.epochs - 1}', mem, targets.shape[0],
                                                                          imgs.shape[-1], *self.meter.get_avg()))
            if callbacks is not None:
                callbacks.run('on_train_batch_end', ni, self.model, imgs, targets, paths, self.plots, self.sync_bn, self.cfg.Dataset.np)
        return items

    def _pseudo_label_hit_rate(self, unlabeled_gt):
        """the progress-bar statistics of ssod_trainer.py:657-673 (tp / fp_cls / fp_loc / pse_num / gt_num).  Without ground truth
        (ssod_hyp.with_gt False, the recipes' setting) they are counts of reliable / uncertain pseudo labels
        (utils/self_supervised_utils.py:587-609): computed on the device from the padded pseudo-label table; the reference's
        meter reads them back with .item(), as it does for the loss items.  With ground truth the reference's own matching
        routine runs on the compacted rows (logging code, host side in the reference too)."""
        t9, valid = self._last_pseudo
        lo = torch.as_tensor(self.compute_un_sup_loss.ignore_thres_low, dtype=torch.float64, device=t9.device)
        hi = torch.as_tensor(self.compute_un_sup_loss.ignore_thres_high, dtype=torch.float64, device=t9.device)
        v = valid.bool()
        bs = self.batch_size // self.WORLD_SIZE
        if getattr(self, "target_with_gt", False):
            if not bool(v.any()):
                return dict(tp=0, fp_cls=0, fp_loc=0, pse_num=0, gt_num=0)
            from utils.self_supervised_utils import check_pseudo_label_with_gt          # the user's tree
            tp, fp_cls, fp_loc, pse, gt = check_pseudo_label_with_gt(
                t9[v].float().cpu(), unlabeled_gt.cpu(), ignore_thres_low=lo.tolist(), ignore_thres_high=hi.tolist(), batch_size=bs)
            return dict(tp=tp, fp_cls=fp_cls, fp_loc=fp_loc, pse_num=pse, gt_num=gt)
        cls = t9[:, 1].long().clamp_(0, lo.numel() - 1)
        conf = t9[:, 6]
        reliable = (v & (conf >= hi[cls])).sum().double() / bs
        uncertain = (v & (conf < hi[cls]) & (conf >= lo[cls])).sum().double() / bs
        n = v.sum().double()
        both = reliable + uncertain
        zero = torch.zeros((), dtype=torch.float64, device=t9.device)
        tp = torch.where(both > 0, reliable / both.clamp_min(1e-300), zero)
        recall = torch.where(n > 0, both * bs / n.clamp_min(1.0), zero)
        # no pseudo label at all: the reference logs zeros (:657-659); the formulas above give zeros there too
        return dict(tp=tp, fp_cls=0, fp_loc=recall, pse_num=both, gt_num=reliable)

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    overlap_teacher = True
    teacher_after = "p3"
    join_teacher_late = True


def hot_path_trainers(ref_trainer=None, ref_ssod_trainer=None, compute_dtype=torch.bfloat16, deterministic=False):
    """(Trainer, SSODTrainer): subclasses of the reference's trainers -- taken from the tree this is called in
    (``trainer.trainer.Trainer`` / ``trainer.ssod_trainer.SSODTrainer``) unless passed explicitly.
    compute_dtype: torch.bfloat16 (default: no loss scaling), torch.float16 (the reference's autocast dtype; ``self.scaler`` is then
    the device-resident GradScaler the reference's ``update_optimizer`` drives unchanged) or torch.float32 (parity mode).
    deterministic: BatchNorm statistics of the 16-bit modes on the reproducible partial-row path (FlatState(deterministic=True))."""
    if compute_dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise ValueError(f"compute_dtype {compute_dtype}")
    if ref_trainer is None or ref_ssod_trainer is None:
        from trainer.ssod_trainer import SSODTrainer as ref_ssod_trainer       # noqa: N813  (the user's tree)
        from trainer.trainer import Trainer as ref_trainer                     # noqa: N813
    hot = type("Trainer", (_HotPath, ref_trainer), {"__doc__": "reference Trainer with the MI355X hot path", "ET_COMPUTE_DTYPE": compute_dtype,
                                                      "ET_DETERMINISTIC": bool(deterministic)})
    ssod = type("SSODTrainer", (_SsodHotPath, ref_ssod_trainer), {"__doc__": "reference SSODTrainer with the MI355X hot path",
                                                                   "ET_COMPUTE_DTYPE": compute_dtype, "ET_DETERMINISTIC": bool(deterministic)})
    return hot, ssod
