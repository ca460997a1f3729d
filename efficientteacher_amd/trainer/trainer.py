"""Supervised trainer core (host-side mirror of reference trainer/trainer.py: build_model :125,
build_optimizer :193, build_ddp_model :308, update_optimizer :381, the train_in_epoch body :406-443).

Only the per-iteration hot path is mirrored: data loading, validation, checkpoints and loggers are the
reference's own host code (out of scope) and plug in around ``train_step``.
Mixed precision: the reference autocasts to fp16 and scales the loss (trainer.py:248,348); here the default
performance mode is bf16 storage with fp32 accumulation and fp32 master weights, which needs no loss
scaling (``self.scaler`` disabled = identity); ``model.set_compute_dtype(torch.float16)`` before
``build_optimizer`` selects the reference's own arithmetic, and ``self.scaler`` is then a live
``optim.DeviceGradScaler`` driven exactly where the reference drives its GradScaler (trainer.py:399-401).
"""
import logging

import numpy as np
import torch
from torch.optim import lr_scheduler

from ..models.loss import ComputeLoss
from ..optim import DeviceGradScaler, FlatSGD
from ..parallel import FlatDataParallel
from ..utils.torch_utils import ModelEMA

LOGGER = logging.getLogger(__name__)


class Trainer:
    MODEL_MODULE = "efficientteacher_amd.models.detector.yolo"

    def __init__(self, cfg, device, callbacks=None, LOCAL_RANK=-1, RANK=-1, WORLD_SIZE=1, nb=1000, amp_dtype=None):
        """amp_dtype: compute dtype on a GPU -- None / torch.bfloat16 (default performance mode), torch.float16 (the reference's
        autocast dtype: enables the loss scaler), torch.float32 (parity mode)"""
        self.cfg = cfg
        self._amp_dtype_arg = amp_dtype
        self.set_env(cfg, device, LOCAL_RANK, RANK, WORLD_SIZE, callbacks, nb)
        self.build_model(cfg, device)
        self.build_optimizer(cfg)
        self.build_ddp_model(cfg, device)

    # ---- reference trainer.py:253-306 (the parts the step needs) ------------------------------------------
    def set_env(self, cfg, device, LOCAL_RANK, RANK, WORLD_SIZE, callbacks, nb=1000):
        self.device = torch.device(device)
        self.cuda = self.device.type != 'cpu'
        self.LOCAL_RANK, self.RANK, self.WORLD_SIZE = LOCAL_RANK, RANK, WORLD_SIZE
        self.callbacks = callbacks
        self.epochs = cfg.epochs
        self.epoch = 0
        self.start_epoch = 0
        self.batch_size = cfg.Dataset.batch_size
        self.imgsz = cfg.Dataset.img_size
        self.norm_scale = cfg.Dataset.norm_scale if 'norm_scale' in cfg.Dataset else 255.0
        self.warmup_epochs = cfg.hyp.warmup_epochs
        self.warmup_momentum = cfg.hyp.warmup_momentum
        self.warmup_bias_lr = cfg.hyp.warmup_bias_lr
        self.momentum = cfg.hyp.momentum
        self.nb = nb                                   # batches per epoch (set by whoever owns the loader)
        self.last_opt_step = -1
        self.amp_dtype = (getattr(self, "_amp_dtype_arg", None) or torch.bfloat16) if self.cuda else torch.float32
        self.model_type = 'yolov5'
        self.sync_bn = False
        if cfg.sync_bn:
            raise NotImplementedError("SyncBatchNorm is off in every shipped config (SURVEY.md 2a)")

    def _model_class(self):
        import importlib
        return importlib.import_module(self.MODEL_MODULE).Model

    def build_model(self, cfg, device):
        self.model = self._model_class()(cfg).to(device)
        self.model.set_compute_dtype(self.amp_dtype)
        ckpt = None
        if str(cfg.weights).endswith('.pt'):       # trainer.py:127-144: intersect by name/shape, anchors excluded unless resuming
            from ..utils.checkpoint import load_reference_checkpoint
            ckpt = load_reference_checkpoint(cfg.weights, map_location="cpu")
            msd = self.model.state_dict()
            exclude = ['anchor'] if not cfg.resume else []
            csd = {k: v for k, v in ckpt["model"].items() if k in msd and not any(x in k for x in exclude)
                   and tuple(v.shape) == tuple(msd[k].shape)}
            self.model.load_state_dict(csd, strict=False)      # marks the flat arenas changed (bf16 shadow, transposes)
            LOGGER.info(f'Transferred {len(csd)}/{len(msd)} items from {cfg.weights}')
        for _, v in self.model.named_parameters():
            v.requires_grad = True
        self.ema = ModelEMA(self.model)                        # built AFTER the load: the teacher starts from the loaded weights
        if ckpt is not None:
            if ckpt.get("ema"):
                esd = self.ema.ema.state_dict()
                self.ema.ema.load_state_dict({k: v for k, v in ckpt["ema"].items() if k in esd and tuple(v.shape) == tuple(esd[k].shape)},
                                             strict=False)
            if ckpt.get("updates") is not None:
                self.ema.updates = ckpt["updates"]
            if ckpt.get("epoch") is not None:
                self.start_epoch = ckpt["epoch"] + 1
                self.epoch = self.start_epoch
        self._ckpt = ckpt

    def build_optimizer(self, cfg):
        inner = self.model.module if isinstance(self.model, FlatDataParallel) else self.model
        # trainer.py:248 GradScaler(enabled=cuda): live in fp16 mode only (bf16 / fp32 need no loss scaling)
        self.scaler = DeviceGradScaler(self.device, enabled=getattr(inner, "_compute_dtype", None) == torch.float16)
        nbs = 64  # nominal batch size
        self.accumulate = max(round(nbs / self.batch_size), 1)
        weight_decay = cfg.hyp.weight_decay * self.batch_size * self.accumulate / nbs
        if cfg.adam:                                   # trainer.py:210-212
            from ..optim import FlatAdamW
            self.optimizer = FlatAdamW(self.model, lr=cfg.hyp.lr0, betas=(cfg.hyp.momentum, 0.999), weight_decay=weight_decay)
        else:
            self.optimizer = FlatSGD(self.model, lr=cfg.hyp.lr0, momentum=cfg.hyp.momentum, nesterov=True,
                                     weight_decay=weight_decay)
        if cfg.linear_lr:
            self.lf = lambda x: (1 - x / (self.epochs - 1)) * (1.0 - cfg.hyp.lrf) + cfg.hyp.lrf
        else:
            import math
            self.lf = lambda x: ((1 - math.cos(x * math.pi / self.epochs)) / 2) * (cfg.hyp.lrf - 1) + 1
        self.scheduler = lr_scheduler.LambdaLR(self.optimizer, lr_lambda=self.lf)
        self.scheduler.last_epoch = self.epoch - 1
        ck = getattr(self, "_ckpt", None)
        if ck is not None and ck.get("optimizer") is not None:   # trainer.py:249-251
            try:
                self.optimizer.load_state_dict(ck["optimizer"])
            except (ValueError, KeyError, RuntimeError):
                LOGGER.info("checkpoint optimizer state belongs to another optimizer type: starting it fresh")
        # warm-up length (trainer.py:372-376): none at all when hyp.warmup_epochs == 0 (the default of configs/defaults.py)
        if self.warmup_epochs > 0:
            self.nw = max(round(self.warmup_epochs * self.nb), 1000)
            self.nw = min(self.nw, (self.epochs - self.start_epoch) / 2 * self.nb)
        else:
            self.nw = -1

    def build_ddp_model(self, cfg, device):
        from .. import _lib
        if (self.cuda or _lib.is_emulated()) and self.RANK != -1:      # emulated: the world-size-2 gloo tests on CPU
            self.model = FlatDataParallel(self.model)
            self._sync_ema_from_rank0()
        if cfg.Loss.type == 'ComputeTalLoss':          # YOLOv8 head (trainer.py:320-327 dispatches on cfg.Loss.type)
            from ..models.loss import ComputeTalLoss
            self.compute_loss = ComputeTalLoss(self.model, cfg)
        else:
            self.compute_loss = ComputeLoss(self.model, cfg)

    def _sync_ema_from_rank0(self):
        """The EMA model is a copy of the student taken in build_model, i.e. BEFORE the data-parallel wrapper broadcast rank 0's
        parameters (trainer.py:125 vs :313; the reference seeds every rank differently, trainer.py:294).  The student is made
        identical on all ranks by the broadcast; the EMA teacher -- which the SSOD step runs on every rank -- is made
        identical here the same way (a no-op when the ranks started from one checkpoint, as the reference's recipes do)."""
        import torch.distributed as dist
        m = self.model
        if not (isinstance(m, FlatDataParallel) and m.active):
            return
        for e in (getattr(self, "ema", None), getattr(self, "semi_ema", None)):
            if e is None:
                continue
            f = e.ema.flat_state()
            dist.broadcast(f.params, 0, group=m.pg)
            dist.broadcast(f.buffers, 0, group=m.pg)
            f.mark_weights_changed()

    # ---- the step ------------------------------------------------------------------------------------------
    def _warmup(self, ni, accumulate_target):
        if ni <= self.nw:
            xi = [0, self.nw]
            self.accumulate = max(1, np.interp(ni, xi, [1, accumulate_target]).round())
            for j, x in enumerate(self.optimizer.param_groups):
                # (sic) group 2 = BN weights gets warmup_bias_lr, exactly as the reference (trainer.py:393)
                x['lr'] = np.interp(ni, xi, [self.warmup_bias_lr if j == 2 else 0.0, x['initial_lr'] * self.lf(self.epoch)])
                if 'momentum' in x:
                    x['momentum'] = np.interp(ni, xi, [self.warmup_momentum, self.momentum])

    def update_optimizer(self, loss, ni):
        self.scaler.scale(loss).backward()                     # trainer.py:399
        if isinstance(self.model, FlatDataParallel):
            self.model.reduce_gradients()
        self.accumulate = max(round(64 / self.batch_size), 1)
        self._warmup(ni, 64 / self.batch_size)
        if ni - self.last_opt_step >= self.accumulate:
            self.scaler.step(self.optimizer)                   # trainer.py:400-401 (skips on inf / nan, on the device)
            self.scaler.update()
            self.optimizer.zero_grad()
            if self.ema:
                self.ema.update(self.model)
            self.last_opt_step = ni

    def train_step(self, imgs, targets, ni):
        """Body of train_in_epoch (trainer.py:411-430) for one batch: imgs uint8/float NCHW, targets (n,6)."""
        imgs = imgs.to(self.device, non_blocking=True)
        if imgs.dtype == torch.uint8 and self.cuda:
            # the loaders' uint8 batch goes to the model as it is: the division by norm_scale (trainer.py:414) happens inside
            # the input pack kernel (et_pack_input_u8), bit-equal to the float division
            inner = self.model.module if isinstance(self.model, FlatDataParallel) else self.model
            inner.input_norm_scale = float(self.norm_scale)
        else:
            imgs = imgs.float() / self.norm_scale
        pred = self.model(imgs)
        loss, loss_items = self.compute_loss(pred, targets.to(self.device))
        if self.RANK != -1:
            loss = loss * self.WORLD_SIZE      # gradient averaged between devices in DDP mode
        self.update_optimizer(loss, ni)
        return loss_items
