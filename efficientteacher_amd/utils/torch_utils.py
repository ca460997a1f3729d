"""Host-side mirror of the reference's ``utils/torch_utils.py`` for the hot path: the three EMA
classes (ModelEMA :308, SemiSupModelEMA :344, CosineEMA :381) with their exact decay schedules, and
small helpers.  The per-tensor python loop of the reference's ``update`` is one flat-arena kernel
launch per arena here (et_ema_update, csrc/optim.hip)."""
import math
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def is_parallel(model):
    # Returns True if model is of type DP or DDP (or this package's flat data-parallel wrapper)
    from ..parallel import FlatDataParallel
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel, FlatDataParallel)


def de_parallel(model):
    return model.module if is_parallel(model) else model


def copy_attr(a, b, include=(), exclude=()):
    """public attributes of b -> a (all of them, or only `include`), minus `exclude`; ModelEMA.update_attr uses it"""
    wanted = set(include)
    for name, value in vars(b).items():
        if name.startswith('_') or name in exclude or (wanted and name not in wanted):
            continue
        setattr(a, name, value)


def initialize_weights(model):
    """the BatchNorm constants every reference model is built with (utils/torch_utils.py:162: eps 1e-3, momentum 0.03);
    in-place activations are the fused kernels' business here, not a module flag"""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps, m.momentum = 1e-3, 0.03


def _flat_ema_update(ema_model, model, d, d_dev=None):
    """v = v*d + (1-d)*m for every floating state tensor (parameters AND buffers).  d_dev: device tensor [d, 1-d]
    (graph-replayable launch, the host value `d` is then not used)."""
    src = de_parallel(model)
    fe, fm = ema_model.flat_state(), src.flat_state()
    if d_dev is not None:
        ops.ema_update_dev(fe.params, fm.params, d_dev)
        ops.ema_update_dev(fe.buffers, fm.buffers, d_dev)
    else:
        ops.ema_update(fe.params, fm.params, d)
        ops.ema_update(fe.buffers, fm.buffers, d)
    fe.mark_weights_changed()


class _EMABase:
    d_dev = None          # device [d, 1-d]: set by trainer/graph_step.py, update() then launches the replayable kernels
    capturing = False     # inside a graph capture nothing executes: update() must not advance the host-side schedule

    def _make(self, model):
        self.ema = deepcopy(de_parallel(model)).eval()  # FP32 EMA
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def advance(self):
        """host side of one update: bump the counters, return the decay this update uses"""
        raise NotImplementedError

    def update(self, model):
        with torch.no_grad():
            d = None if self.capturing else self.advance()
            _flat_ema_update(self.ema, model, d, self.d_dev if self.capturing else None)

    def update_attr(self, model, include=(), exclude=('process_group', 'reducer')):
        copy_attr(self.ema, model, include, exclude)


class ModelEMA(_EMABase):
    """decay ramp d = decay * (1 - exp(-updates / 2000))  (reference :324)."""

    def __init__(self, model, decay=0.9999, updates=0):
        self._make(model)
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))

    def advance(self):
        self.updates += 1
        return self.decay(self.updates)


class SemiSupModelEMA(_EMABase):
    """constant decay (reference :358)."""

    def __init__(self, model, decay=0.99, updates=0):
        self._make(model)
        self.updates = updates
        self.decay = decay

    def advance(self):
        self.updates += 1
        return self.decay


class CosineEMA(_EMABase):
    """decay_start -> decay_end on a cosine over the epochs (reference :391-419)."""

    def __init__(self, model, decay_start=0.99, decay_end=0.9999, total_epoch=0):
        self._make(model)
        self.total_epoch = total_epoch
        self.decay_start = decay_start
        self.decay_end = decay_end
        self.decay = decay_start
        self.updates = 0

    def advance(self):
        return self.decay

    def update_decay(self, cur_epoch):
        self.decay = self.decay_end - (self.decay_end - self.decay_start) * \
            (np.cos(np.pi * cur_epoch / self.total_epoch) + 1) / 2
