"""Flat HBM arenas for every floating tensor of a detector (parameters, gradients, momentum, BN
running statistics) -- the memory layout the MI355X step is built around.

Why: the reference walks ~620 state tensors in Python for every EMA update (utils/torch_utils.py:335),
lets torch.optim.SGD launch per-tensor kernels and lets DDP re-bucket 191.8 MB of gradients each step.
With 288 GB of HBM per GPU the whole model state is laid out once, contiguously, grouped exactly like
the reference's optimizer groups (trainer/trainer.py:199-217):

    params  = [ g_b : BN biases | conv biases ][ g_w : conv weights (decay) ][ g_bnw : BN weights ]
    buffers = [ BN running_mean | BN running_var | other float buffers (Detect.anchors) ]

so that one EMA launch, three SGD launches and a handful of large RCCL all-reduces cover everything.
``nn.Parameter``s / buffers keep their reference names and logical shapes: each is a strided *view*
into an arena (conv weights are stored [CoutP][KH][KW][CinP], channel counts padded to 8 -- the
layout the implicit-GEMM kernels read -- and exposed as the usual (Cout, Cin, KH, KW) view).
"""
import weakref

import torch
import torch.nn as nn

from . import ops

ALIGN = 16  # floats (64 B)


def _pad8(n):
    return (n + 7) // 8 * 8


def _round(n, a=ALIGN):
    return (n + a - 1) // a * a


class ConvSlot:
    """Kernel-side view of one nn.Conv2d: padded weight / grad / low-precision shadow, bias."""
    __slots__ = ("cout", "cin", "coutp", "cinp", "k", "stride", "pad", "w", "gw", "w_lp", "bias", "gbias", "index",
                 "wT_lp", "flat")

    def transposed(self):
        """(Cin, KH, KW, Cout) copy of the compute-precision weight: the dgrad operand.  All layers are
        re-transposed by ONE launch the first time any of them is needed after a weight update."""
        f = self.flat()
        f.ensure_transposed()
        return self.wT_lp


BN_SHARDS = 16           # ET_BN_SHARDS of include/et_hip.h
BN_SHARD_MAX_C = 1024    # et_bn_act_fwd_sharded / et_bn_act_bwd_sharded


class BnSlot:
    __slots__ = ("c", "gamma", "beta", "ggamma", "gbeta", "rmean", "rvar", "eps", "momentum", "aff_off",
                 "sh_fwd", "sh_bwd", "sh_ld", "sh_view", "gen", "fwd_gen", "bwd_gen")

    # Sharded statistics (16-bit training modes; et_hip.h, et_conv2d_fwd stats_ld > 0): every BN layer owns channel range
    # [aff_off, aff_off + c) of two zero-initialised [BN_SHARDS][2][bn_total] accumulators, one for the forward sums and one for the
    # backward sums.  FlatState.prepare_forward zeroes both arenas with ONE memset per training forward and bumps the generation; a slot
    # that is used a second time inside one generation (a module applied twice, two backward passes through one forward) zeroes its own
    # range first -- correct in every call pattern, one launch per layer cheaper in the usual one.
    def _acquire(self, which):
        if self.sh_ld == 0:
            return None
        used = self.fwd_gen if which == 0 else self.bwd_gen
        if used == self.gen[0]:
            self.sh_view[which].zero_()
        if which == 0:
            self.fwd_gen = self.gen[0]
        else:
            self.bwd_gen = self.gen[0]
        return (self.sh_fwd if which == 0 else self.sh_bwd), self.sh_ld

    def acquire_fwd(self, n_channels=None):
        """(tensor at this layer's channel 0, ld) of the zeroed forward accumulator, or None (fp32 mode / too wide)"""
        return self._acquire(0)

    def acquire_bwd(self):
        return self._acquire(1)


class FlatState:
    def __init__(self, model, compute_dtype=torch.float32, deterministic=False):
        """deterministic: BatchNorm statistics of the 16-bit modes through the partial-row form (et_conv2d_fwd stats_ld == 0 +
        fp64 finalize: the fp32 parity mode's path, bit-reproducible sums, one more launch per layer and pass) instead of the sharded
        fp32 accumulators, whose last bits depend on the arrival order of hardware atomics.  Selected by Model.set_deterministic() /
        cfg.Model.deterministic_bn / hot_path_trainers(deterministic=True); ~+0.6 ms on the YOLOv5l 32 + 32 step."""
        self.compute_dtype = compute_dtype
        self.deterministic = bool(deterministic)
        dev = next(model.parameters()).device
        convs = [m for m in model.modules() if isinstance(m, nn.Conv2d)]
        bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
        for m in convs:
            assert m.groups == 1 and m.kernel_size[0] == m.kernel_size[1] and m.stride[0] == m.stride[1]
        # ---- layout -------------------------------------------------------------------------------------
        off = 0
        self.bn_off = []                      # per BN: offset inside each of the 4 BN-channel segments
        bn_total = 0
        for b in bns:
            self.bn_off.append(bn_total)
            bn_total += _round(b.num_features)
        self.bn_total = bn_total
        seg = {}
        seg["bn_bias"] = (off, bn_total); off += bn_total
        cb = []
        for m in convs:
            if m.bias is not None:
                cb.append((m, off)); off += _round(_pad8(m.out_channels))
        seg["g_b"] = (0, off)
        w_begin = off
        cw = []
        for m in convs:
            n = _pad8(m.out_channels) * m.kernel_size[0] * m.kernel_size[1] * _pad8(m.in_channels)
            cw.append((m, off, n)); off += _round(n)
        seg["g_w"] = (w_begin, off - w_begin)
        seg["bn_weight"] = (off, bn_total); seg["g_bnw"] = (off, bn_total); off += bn_total
        self.segments = seg
        self.n_params = off
        # ---- allocate + copy the current values in ---------------------------------------------------------
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        other = [(n, b) for n, b in model.named_buffers() if b.dtype.is_floating_point
                 and not n.endswith("running_mean") and not n.endswith("running_var")]
        nb = 2 * bn_total + sum(_round(b.numel()) for _, b in other)
        self.buffers = torch.zeros(nb, dtype=torch.float32, device=dev)
        self.rm_off, self.rv_off = 0, bn_total
        self.w_range = seg["g_w"]
        self.shadow = None
        if compute_dtype != torch.float32:
            self.shadow = torch.zeros(self.w_range[1], dtype=compute_dtype, device=dev)

        def repoint(param, view, gview):
            with torch.no_grad():
                view.copy_(param.data.to(torch.float32))
            param.data = view
            param.grad = gview

        # transposed copy of the compute-precision weights (dgrad operand), same layout as the weight segment
        self.shadow_T = torch.zeros(self.w_range[1], dtype=compute_dtype, device=dev)
        self.w_version, self.wT_version = 1, 0
        rows = []
        self.conv_slots, self.bn_slots = {}, {}
        bias_off = dict((id(m), o) for m, o in cb)
        for idx, (m, o, n) in enumerate(cw):
            s = ConvSlot()
            s.index = idx
            s.cout, s.cin, s.k = m.out_channels, m.in_channels, m.kernel_size[0]
            s.coutp, s.cinp = _pad8(s.cout), _pad8(s.cin)
            s.stride, s.pad = m.stride[0], m.padding[0]
            shape = (s.coutp, s.k, s.k, s.cinp)
            s.w = self.params[o:o + n].view(shape)
            s.gw = self.grads[o:o + n].view(shape)
            repoint(m.weight, s.w.permute(0, 3, 1, 2)[:s.cout, :s.cin], s.gw.permute(0, 3, 1, 2)[:s.cout, :s.cin])
            so = o - self.w_range[0]
            if self.shadow is not None:
                s.w_lp = self.shadow[so:so + n].view(shape)
            else:
                s.w_lp = s.w
            s.wT_lp = self.shadow_T[so:so + n].view(s.cinp, s.k, s.k, s.coutp)
            s.flat = weakref.ref(self)
            rows.append((so, s.coutp, s.k * s.k, s.cinp))
            s.bias = s.gbias = None
            if m.bias is not None:
                bo = bias_off[id(m)]
                s.bias = self.params[bo:bo + s.coutp]
                s.gbias = self.grads[bo:bo + s.coutp]
                repoint(m.bias, s.bias[:s.cout], s.gbias[:s.cout])
            self.conv_slots[id(m)] = s
            m._et_slot = s
            m._et_flat_ref = weakref.ref(self)
        self.wT_table = torch.tensor(sorted(rows), dtype=torch.int32, device=dev).reshape(-1, 4)
        bw0 = seg["bn_weight"][0]
        self._bn_gen = [0]
        self.bn_shards = (torch.zeros((2, BN_SHARDS, 2, max(bn_total, 1)), dtype=torch.float32, device=dev)
                          if compute_dtype != torch.float32 and not self.deterministic else None)
        for b, o in zip(bns, self.bn_off):
            c = b.num_features
            s = BnSlot()
            s.c, s.eps, s.momentum, s.aff_off = c, b.eps, b.momentum, o
            s.gen, s.fwd_gen, s.bwd_gen = self._bn_gen, -1, -1
            if self.bn_shards is not None and c <= BN_SHARD_MAX_C:
                s.sh_ld = bn_total
                s.sh_fwd = self.bn_shards[0].view(-1)[o:]
                s.sh_bwd = self.bn_shards[1].view(-1)[o:]
                s.sh_view = (self.bn_shards[0][:, :, o:o + c], self.bn_shards[1][:, :, o:o + c])
            else:
                s.sh_ld, s.sh_fwd, s.sh_bwd, s.sh_view = 0, None, None, None
            s.gamma = self.params[bw0 + o:bw0 + o + c]; s.ggamma = self.grads[bw0 + o:bw0 + o + c]
            s.beta = self.params[o:o + c]; s.gbeta = self.grads[o:o + c]
            repoint(b.weight, s.gamma, s.ggamma)
            repoint(b.bias, s.beta, s.gbeta)
            s.rmean = self.buffers[self.rm_off + o:self.rm_off + o + c]
            s.rvar = self.buffers[self.rv_off + o:self.rv_off + o + c]
            with torch.no_grad():
                s.rmean.copy_(b.running_mean); s.rvar.copy_(b.running_var)
            b.running_mean = s.rmean      # registered buffers: assignment keeps them registered
            b.running_var = s.rvar
            self.bn_slots[id(b)] = s
            b._et_slot = s
        # num_batches_tracked of every BatchNorm as views of ONE int64 vector: a training forward bumps all of
        # them with a single launch (was 101 one-element kernels per step)
        self.nbt = torch.zeros(max(1, len(bns)), dtype=torch.int64, device=dev)
        self._bns = bns
        self.bulk_nbt = False
        for i, b in enumerate(bns):
            if b.num_batches_tracked is not None:
                with torch.no_grad():
                    self.nbt[i] = b.num_batches_tracked.to(dev)
                b.num_batches_tracked = self.nbt[i]
        bo = 2 * bn_total
        owner = dict(model.named_modules())
        for name, buf in other:
            n = buf.numel()
            view = self.buffers[bo:bo + n].view(buf.shape)
            with torch.no_grad():
                view.copy_(buf)
            mod_name, _, leaf = name.rpartition(".")
            setattr(owner[mod_name], leaf, view)
            bo += _round(n)
        # eval-mode (teacher) BN affine for ALL layers in one launch: scale/shift arenas
        self.eval_scale = torch.zeros(bn_total, dtype=torch.float32, device=dev)
        self.eval_shift = torch.zeros(bn_total, dtype=torch.float32, device=dev)
        # the padding lanes of running_var must not be 0 (1/sqrt(0 + eps) is finite, fine) -- nothing to do
        self.momentum_buf = None
        self.weights_dirty = False
        self.zero_gen = 0
        self.sync_shadow()

    # ---- maintenance ------------------------------------------------------------------------------------------
    def sync_shadow(self):
        """Refresh the low-precision weight copy from the fp32 master (after load_state_dict etc.)."""
        if self.shadow is not None:
            o, n = self.w_range
            ops.cast_f32_to_lp(self.params[o:o + n], self.shadow)

    def refresh_eval_affine(self):
        """scale = g / sqrt(running_var + eps), shift = b - running_mean*scale for every BN channel."""
        eps = next(iter(self.bn_slots.values())).eps if self.bn_slots else 1e-3
        bw0 = self.segments["bn_weight"][0]
        ops.bn_eval_affine_into(self.params[bw0:bw0 + self.bn_total], self.params[0:self.bn_total],
                                self.buffers[self.rm_off:self.rm_off + self.bn_total],
                                self.buffers[self.rv_off:self.rv_off + self.bn_total], eps, self.eval_scale,
                                self.eval_shift)

    def mark_weights_changed(self):
        """The fp32 master changed outside the fused SGD (EMA update, load_state_dict, torch optimizers)."""
        self.weights_dirty = True
        self.w_version += 1

    def ensure_transposed(self):
        if self.wT_version != self.w_version:
            src = self.shadow if self.shadow is not None else self.params[self.w_range[0]:self.w_range[0] + self.w_range[1]]
            ops.weight_transpose_all(src, self.shadow_T, self.wT_table, self.w_range[1])
            self.wT_version = self.w_version

    def prepare_forward(self, training):
        """Called by Model.forward: bring the bf16 shadow / eval-mode BN affine up to date if needed."""
        if training:
            ops.WGRAD_QUEUE.reset()      # nothing of an earlier (failed) backward lingers; side stream joined
            if self.bn_shards is not None:
                self.bn_shards.zero_()
                self._bn_gen[0] += 1
        if self.weights_dirty:
            self.sync_shadow()
            self.weights_dirty = False
        if not training:
            self.refresh_eval_affine()
        # reference: every BatchNorm in training mode increments its counter once per forward
        self.bulk_nbt = bool(training) and all(b.training for b in self._bns)
        if self.bulk_nbt:
            self.nbt.add_(1)

    def zero_grad(self):
        self.grads.zero_()
        self.zero_gen += 1               # parallel.FlatDataParallel: "the arena has been zeroed since ..."

    def group_ranges(self):
        """[(offset, numel)] of the optimizer groups in the reference's order [biases, weights, BN weights]."""
        return [self.segments["g_b"], self.segments["g_w"], self.segments["g_bnw"]]
