"""TEST INFRASTRUCTURE -- CPU restatement of the per-step state updates.

  * EMA: ``ModelEMA.update`` / ``SemiSupModelEMA.update`` / ``CosineEMA.update``
    (utils/torch_utils.py:330-338, 366-375, 406-416): for every *floating* state
    tensor ``v = v*d + (1-d)*m`` evaluated as ``v *= d; v += (1-d)*m`` in fp32, with
    d = decay*(1-exp(-updates/2000)) (ModelEMA :324), const (SemiSup :358) or cosine
    per epoch (``update_decay`` :418-419).
  * SGD: ``torch.optim.SGD(momentum, nesterov=True)`` as built at
    trainer/trainer.py:215-223 (3 groups; weight decay on group 1 only):
        g = g + wd*p ; buf = mu*buf + g (buf=g on first step) ; g = g + mu*buf ; p -= lr*g
"""
import math

import numpy as np


def ema_decay_ramp(updates, decay=0.9999):
    return decay * (1 - math.exp(-updates / 2000))


def cosine_decay(cur_epoch, total_epoch, decay_start, decay_end=0.9999):
    return decay_end - (decay_end - decay_start) * (np.cos(np.pi * cur_epoch / total_epoch) + 1) / 2


def ema_update(v, m, d):
    """fp32 arrays; returns the new EMA value with the reference's op order."""
    v = np.asarray(v, np.float32) * np.float32(d)
    return v + np.float32(1. - d) * np.asarray(m, np.float32)


def sgd_nesterov(p, g, buf, lr, momentum, weight_decay, first_step):
    p, g = np.asarray(p, np.float32), np.asarray(g, np.float32)
    if weight_decay != 0:
        g = g + np.float32(weight_decay) * p
    if first_step:
        buf = g.copy()
    else:
        buf = np.asarray(buf, np.float32) * np.float32(momentum) + g
    g = g + np.float32(momentum) * buf
    return p - np.float32(lr) * g, buf
