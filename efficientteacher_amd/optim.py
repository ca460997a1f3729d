"""SGD(momentum, nesterov) over the flat parameter arena: three launches (one per optimizer group of
reference trainer/trainer.py:199-217: [biases | weights with decay | BN weights]) instead of one
kernel chain per tensor.  It is a ``torch.optim.Optimizer`` so the reference's schedulers
(LambdaLR / MultiStepLR, trainer.py:242, ssod_trainer.py:90) and its warm-up code, which edits
``optimizer.param_groups[j]['lr' | 'momentum']`` (trainer.py:391-395), work unchanged.
The bf16 shadow copy of the conv weights is refreshed by the same kernel.
"""
import torch

from . import ops


class DeviceGradScaler:
    """torch.cuda.amp.GradScaler as the reference drives it (trainer/trainer.py:248 ``GradScaler(enabled=cuda)``, :399
    ``scaler.scale(loss).backward()``, :400-401 ``scaler.step(optimizer); scaler.update()``; ssod_trainer.py:595,625) for the fp16
    compute mode -- with its state {scale, 1/scale, found_inf, growth_tracker} in DEVICE memory: ``step`` checks the (scaled, fp32)
    gradient arena for inf / nan with one pass (et_scaler_check) and the optimizer kernels themselves skip the update or unscale
    the gradient as they read it, so that a step never synchronises with the host (torch's ``scaler.step`` calls
    ``found_inf.item()``) and can be captured into the step graph.  Defaults are torch's: init 65536, growth 2.0 every 2000 clean
    steps, backoff 0.5.  ``enabled=False`` (fp32 parity mode, bf16 performance mode: no loss scaling needed) makes every method
    the identity."""

    def __init__(self, device, enabled=True, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.enabled = bool(enabled)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = torch.tensor([init_scale, 1.0 / init_scale, 0.0, 0.0], dtype=torch.float32, device=device) if self.enabled else None

    def scale(self, loss):
        return loss * self.state[0] if self.enabled else loss

    def step(self, optimizer):
        """found_inf over the gradient arena, then the optimizer's (possibly skipped) update"""
        if not self.enabled:
            return optimizer.step()
        ops.WGRAD_QUEUE.join()                   # the deferred weight-gradient launches are in the arena before it is scanned
        ops.scaler_check(optimizer.flat.grads, self.state)
        return optimizer.step(scaler=self.state)

    def update(self):
        if self.enabled:
            ops.scaler_update(self.state, self.growth_factor, self.backoff_factor, self.growth_interval)

    def get_scale(self):
        """host read (synchronises): logging / tests only"""
        return float(self.state[0]) if self.enabled else 1.0

    def state_dict(self):
        return dict(state=self.state.detach().cpu().clone()) if self.enabled else {}

    def load_state_dict(self, sd):
        if self.enabled and "state" in sd:
            self.state.copy_(sd["state"].to(self.state.device))


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, momentum=0.937, nesterov=True, weight_decay=0.0):
        if not nesterov:
            raise NotImplementedError("the reference always builds SGD(nesterov=True) (trainer.py:215)")
        self.flat = model.flat_state()
        g_bnw, g_w, g_b = [], [], []
        import torch.nn as nn
        for v in model.modules():   # same walk as the reference => same groups
            if hasattr(v, 'bias') and isinstance(v.bias, nn.Parameter):
                g_b.append(v.bias)
            if isinstance(v, nn.BatchNorm2d):
                g_bnw.append(v.weight)
            elif hasattr(v, 'weight') and isinstance(v.weight, nn.Parameter):
                g_w.append(v.weight)
        groups = [dict(params=g_b), dict(params=g_w, weight_decay=weight_decay), dict(params=g_bnw)]
        super().__init__(groups, dict(lr=lr, momentum=momentum, nesterov=True, weight_decay=0.0))
        for g, r in zip(self.param_groups, self.flat.group_ranges()):
            g['range'] = r
            g.setdefault('initial_lr', lr)
        self._model = model
        self.momentum_buf = torch.zeros_like(self.flat.params)
        self.first = True
        # graph-replayable form: per-group [lr, momentum, weight_decay, inv_scale] in device memory (trainer/graph_step.py)
        self.hp_dev = None
        self.capturing = False       # only a graph capture launches the device-hyper-parameter kernels

    # the momentum arena and the first-step flag live outside Optimizer.state: carry them through checkpoints
    def state_dict(self):
        sd = super().state_dict()
        sd["flat_momentum"] = self.momentum_buf.detach().cpu().clone()
        sd["flat_first"] = bool(self.first)
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        mom, first = sd.pop("flat_momentum", None), sd.pop("flat_first", None)
        if mom is None:
            raise ValueError("not a FlatSGD state_dict (no flat momentum arena)")
        if mom.numel() != self.momentum_buf.numel():
            raise ValueError("flat momentum arena of another model layout")
        ranges = [g["range"] for g in self.param_groups]
        super().load_state_dict(sd)
        for g, r in zip(self.param_groups, ranges):        # 'range' is layout, not state
            g["range"] = r
        self.momentum_buf.copy_(mom.to(self.momentum_buf.device))
        self.first = bool(first)

    @torch.no_grad()
    def step(self, closure=None, inv_scale=1.0, scaler=None):
        """scaler: DeviceGradScaler.state (fp16 mode) -- the kernels skip the update when its found_inf flag is set and multiply
        the gradient by its 1/scale otherwise"""
        f = self.flat
        if f is not self._model.flat_state():
            raise RuntimeError("the model's arenas were rebuilt (model.to() / set_compute_dtype() / _apply) after this optimizer "
                               "was created: it would update orphaned buffers -- build the optimizer after the last such call")
        # deferred weight-gradient launches must be in the arena (and their side stream joined) before it is read
        assert not ops.WGRAD_QUEUE.pending, "weight gradients still queued: backward() did not finish"
        ops.WGRAD_QUEUE.join()
        for j, g in enumerate(self.param_groups):
            o, n = g['range']
            shadow = f.shadow if (f.shadow is not None and (o, n) == tuple(f.w_range)) else None
            if self.capturing and self.hp_dev is not None:
                ops.sgd_nesterov_dev(f.params[o:o + n], f.grads[o:o + n], self.momentum_buf[o:o + n], shadow, self.hp_dev[j],
                                     self.first, scaler)
            else:
                ops.sgd_nesterov(f.params[o:o + n], f.grads[o:o + n], self.momentum_buf[o:o + n], shadow,
                                 g['lr'], g['momentum'], g['weight_decay'], self.first, inv_scale, scaler)
        self.first = False
        f.w_version += 1                 # the compute-precision shadow changed: transposed copies are stale

    def hp_values(self, inv_scale=1.0):
        """flat list [lr, momentum, weight_decay, inv_scale] per group: what step() would pass by value right now"""
        out = []
        for g in self.param_groups:
            out += [float(g['lr']), float(g['momentum']), float(g['weight_decay']), float(inv_scale)]
        return out

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW over the flat arena, as the reference builds it when ``cfg.adam`` (trainer/trainer.py:212):
    ``AdamW(g_b, lr=lr0, betas=(momentum, 0.999))`` + the decayed-weights group + the BN-weight group.  Kept quirk: AdamW's
    default weight_decay 0.01 therefore applies to the bias and BN-weight groups too (the reference never overrides it)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, default_weight_decay=1e-2):
        import torch.nn as nn
        self.flat = model.flat_state()
        g_bnw, g_w, g_b = [], [], []
        for v in model.modules():
            if hasattr(v, 'bias') and isinstance(v.bias, nn.Parameter):
                g_b.append(v.bias)
            if isinstance(v, nn.BatchNorm2d):
                g_bnw.append(v.weight)
            elif hasattr(v, 'weight') and isinstance(v.weight, nn.Parameter):
                g_w.append(v.weight)
        groups = [dict(params=g_b), dict(params=g_w, weight_decay=weight_decay), dict(params=g_bnw)]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=default_weight_decay))
        for g, r in zip(self.param_groups, self.flat.group_ranges()):
            g['range'] = r
            g.setdefault('initial_lr', lr)
        self._model = model
        self.exp_avg = torch.zeros_like(self.flat.params)
        self.exp_avg_sq = torch.zeros_like(self.flat.params)
        self.steps = 0
        # under a DeviceGradScaler the count lives on the device ({t, 1 - b1^t, 1 - b2^t}, et_adamw_tick): a step the scaler skips
        # (found_inf, known only there) must not advance the bias corrections -- torch's GradScaler.step does not call step() then
        self.tick = None

    def _steps_now(self):
        """the update count (reads the device copy back when the scaler path owns it: checkpoints / tests only)"""
        if self.tick is not None:
            self.steps = int(round(float(self.tick[0].item())))
        return self.steps

    def state_dict(self):
        sd = super().state_dict()
        sd["flat_exp_avg"] = self.exp_avg.detach().cpu().clone()
        sd["flat_exp_avg_sq"] = self.exp_avg_sq.detach().cpu().clone()
        sd["flat_steps"] = int(self._steps_now())
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        m, v, n = sd.pop("flat_exp_avg", None), sd.pop("flat_exp_avg_sq", None), sd.pop("flat_steps", None)
        if m is None or v is None or m.numel() != self.exp_avg.numel():
            raise ValueError("not a FlatAdamW state_dict of this model layout")
        ranges = [g["range"] for g in self.param_groups]
        super().load_state_dict(sd)
        for g, r in zip(self.param_groups, ranges):
            g["range"] = r
        self.exp_avg.copy_(m.to(self.exp_avg.device)); self.exp_avg_sq.copy_(v.to(self.exp_avg_sq.device))
        self.steps = int(n)
        self.tick = None

    @torch.no_grad()
    def step(self, closure=None, inv_scale=1.0, scaler=None):
        f = self.flat
        if f is not self._model.flat_state():
            raise RuntimeError("the model's arenas were rebuilt after this optimizer was created")
        assert not ops.WGRAD_QUEUE.pending, "weight gradients still queued: backward() did not finish"
        ops.WGRAD_QUEUE.join()
        if scaler is not None:
            b1, b2 = self.param_groups[0]['betas']
            if any(tuple(g['betas']) != (b1, b2) for g in self.param_groups):
                raise NotImplementedError("per-group betas under the loss scaler (the reference builds one pair, trainer.py:212)")
            if self.tick is None:
                t = float(self.steps)
                self.tick = torch.tensor([t, 1.0 - b1 ** t, 1.0 - b2 ** t], dtype=torch.float64, device=f.params.device)
            ops.adamw_tick(self.tick, b1, b2, scaler)
        else:
            self.steps = self._steps_now() + 1
            self.tick = None
        for g in self.param_groups:
            o, n = g['range']
            shadow = f.shadow if (f.shadow is not None and (o, n) == tuple(f.w_range)) else None
            if scaler is not None:
                ops.adamw_dev(f.params[o:o + n], f.grads[o:o + n], self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n], shadow, g['lr'],
                              g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.tick, inv_scale, scaler)
            else:
                ops.adamw(f.params[o:o + n], f.grads[o:o + n], self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n], shadow, g['lr'],
                          g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.steps, inv_scale, scaler)
        f.w_version += 1

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()
