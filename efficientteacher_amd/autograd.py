"""torch.autograd glue: one Function per fused block of the YOLOv5 graph.  Each forward/backward is a
short sequence of kernel launches (efficientteacher_amd/ops.py); parameter gradients are accumulated
by the kernels straight into the flat gradient arena (FlatState.grads), so backward returns ``None``
for them and autograd only carries activation gradients.
"""
import torch
from torch.autograd import Function

from . import ops

ACT_CODE = {"silu": ops.ACT_SILU, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}

# the reduce pass of BatchNorm backward inside the epilogue of the dgrad that produces its input gradient
# (et_conv2d_dgrad_bn).  Constants since r05 (were the env knobs ET_FUSE_BN_BWD / ET_FUSE_BN_BWD_K; every arm has an A/B file):
FUSE_BN_BWD = True
# which dgrads carry those sums: bit 1 = the 1x1 layers, bit 2 = the k > 1 layers.  Measured on the YOLOv5l SSOD step, same box,
# alternating (profiles/r03_fuse_bn_bwd_by_kernel_size_ab.txt): 3 / 1 / 2 / 0 = 54.70 / 54.78 / 55.00 / 54.74 ms -- the fusion moves the
# reduce pass's time between the BatchNorm family and the dgrads, the step does not change.  Default 1: the HBM-bound 1x1 dgrads carry
# the sums (one y read instead of a dz + y pass), the MFMA-bound 3x3 dgrads keep a pure GEMM epilogue (their launches were 160 us
# with the sums against 125 us without; the separate reduce pass of the same tensor is ~19 us).
# Bit 4 (r04): the k > 1 layers whose dgrad runs on the 128-row row-shift tiles (conv_gemm_rs_kernel: < 256 channels).  Their epilogue
# issues the producer's y reads of a whole slab round up front (conv.hip conv_epilogue_act, EPF) instead of one load + wait per store
# iteration; the register-bound 256x256 tiles cannot afford that prefetch and keep the separate reduce pass.  Measured, alternating,
# two boxes (profiles/r04_stream_full_and_fuse_ab_current_build.txt, r04_knob_combinations_ab.txt): 1 -> 5 = 52.27 / 52.46 -> 52.15 /
# 52.29 ms and 51.37 / 51.43 -> 51.24 / 51.31 ms: BatchNorm family -0.7 ms, gather-GEMMs +0.6 ms, -0.1 ms net in every pair.  Hence 5 (until r06, below).
# (An earlier A/B of this round that showed +0.15 ms had run a STALE library without the prefetch: profiles/r04_fuse_bn_bwd_rs_tiles_ab.txt.)
# r06, re-measured at the final build (buffer-descriptor pieces in the row-shift kernels, weight gradients on their own stream; bench.py --set
# autograd.FUSE_BN_BWD_K=..., profiles/r06_fuse_bn_bwd_by_kernel_ab.txt): 5 -> 1 = 49.53 -> 49.32 ms over 100 steps, 49.65 -> 49.46 over 20 (three / two
# alternations), 7 = +0.06, 0 = +0.13.  The row-shift tiles lose more to the fatter epilogue now than the reduce pass costs beside the weight-gradient
# stream.  Hence 1: only the HBM-bound 1x1 dgrads carry the sums.
FUSE_BN_BWD_K = 1
_RS_DGRAD = {}


def _dgrad_on_rs_tile(cs, x):
    """does the stride-1 dgrad of conv slot `cs` (input x) run on a conv_gemm_rs_kernel tile?  (asked of the library's own selection)"""
    key = (tuple(x.shape), x.dtype, cs.cinp, cs.coutp, cs.k, cs.stride, cs.pad)
    r = _RS_DGRAD.get(key)
    if r is None:
        N, H, W, _ = x.shape
        r = _RS_DGRAD[key] = cs.stride == 1 and ops.kernel_name("dgrad_full", x.dtype, N, H, W, cs.cinp, cs.coutp, cs.k, cs.stride,
                                                               cs.pad).startswith("conv_gemm_rs_kernel")
    return r


def _fuse_into(cs, bn, x=None):
    """bn (a BnBwdSums or None) if the dgrad of conv slot `cs` may carry it, else None (-> the separate reduce pass)"""
    if bn is None:
        return None
    if FUSE_BN_BWD_K & (1 if cs.k == 1 else 2):
        return bn
    if cs.k > 1 and (FUSE_BN_BWD_K & 4) and x is not None and _dgrad_on_rs_tile(cs, x):
        return bn
    return None
# BatchNorm statistics through sharded fp32 accumulators (flat_state.BnSlot): the conv epilogues / the reduce pass ADD their sums, the
# normalise and apply passes fold the shards themselves -- no finalize launch per layer and pass (~190 launches per YOLOv5l step).
# 16-bit training modes only (the slots of an fp32 FlatState have no shards: exact, reproducible partial rows + fp64 finalize).
SHARDED_BN = True


def _bn_train_fwd(x, w_lp, cs_stride, cs_pad, bs, act, residual=None, out=None):
    """y = conv(x, w), z = act(BN_train(y)) (+ residual): (y, z, scale, shift, mean, invstd)"""
    N, IH, IW, Cin = x.shape
    sh = None
    if SHARDED_BN and bs.sh_ld and ops.few_rows("fwd", x.dtype, N, IH, IW, Cin, w_lp.shape[0], w_lp.shape[1], cs_stride, cs_pad):
        sh = bs.acquire_fwd()
    if sh is not None:
        y = ops.conv2d_fwd(x, w_lp, cs_stride, cs_pad, shards=sh)
        N, OH, OW, _ = y.shape
        z, scale, shift, mean, invstd = ops.bn_act_fwd_sharded(y, sh, N * OH * OW, bs.gamma, bs.beta, bs.eps, bs.momentum, bs.rmean,
                                                               bs.rvar, act, residual=residual, out=out)
        return y, z, scale, shift, mean, invstd
    y, stats = ops.conv2d_fwd(x, w_lp, cs_stride, cs_pad, want_stats=True)
    N, OH, OW, _ = y.shape
    scale, shift, mean, invstd = ops.bn_finalize(stats, N * OH * OW, bs.gamma, bs.beta, bs.eps, bs.momentum, bs.rmean, bs.rvar)
    z = ops.bn_act_fwd(y, scale, shift, act, residual=residual, out=out)
    return y, z, scale, shift, mean, invstd


def _bn_sums(y, scale, shift, act, bs):
    """the BnBwdSums a consumer's dgrad fills for the Conv block (y, bs)"""
    return ops.BnBwdSums(y, scale, shift, act, slot=bs if (SHARDED_BN and bs.sh_ld) else None)


def _bn_train_bwd(dz, y, bs, scale, shift, mean, invstd, act, sums=None, out=None):
    """dy of act(BN_train(y)); sums: the BnBwdSums a dgrad may have filled for exactly this dz"""
    part = sums.take(dz) if sums is not None else None
    if SHARDED_BN and bs.sh_ld and (isinstance(part, tuple) or (part is None and ops.few_reduce_rows(y))):
        sh = part if part is not None else bs.acquire_bwd()
        return ops.bn_act_bwd(dz, y, bs.gamma, scale, shift, mean, invstd, act, bs.ggamma, bs.gbeta, out=out, partial=part, shards=sh)
    return ops.bn_act_bwd(dz, y, bs.gamma, scale, shift, mean, invstd, act, bs.ggamma, bs.gbeta, out=out, partial=part)


# two-consumer tensors: the later consumer's backward adds into the earlier one's gradient in its own kernel (GradFork below)
# instead of a torch bf16 add per tensor by autograd (step-neutral, six ATen launches fewer: NOTEBOOK.md round 3)
GRAD_FORK = True

# set by parallel.FlatDataParallel: callable(ConvSlot) invoked right after a layer's wgrad has been
# launched, so that the gradient all-reduce of finished arena chunks overlaps the rest of backward
GRAD_READY_HOOK = None


class GradFork:
    """A tensor with TWO consumers whose gradients would otherwise be added by autograd in a separate (torch) pass.
    ``a, b, fork = GradFork.split(x)``: branch `b`'s consumer runs its backward FIRST (it is the one created later in the forward
    pass) and its gradient is parked here (TapFn); branch `a`'s consumer -- a conv block or the upsample-concat -- then ADDS
    its own gradient into that buffer inside its kernel (dgrad epilogue `accumulate`, et_upsample2x_bwd `accumulate`) and
    marks the fork merged; ForkFn.backward returns the merged buffer as the gradient of x.  Any other order, layout or dtype
    falls back to the ordinary sum."""

    def __init__(self):
        self.gb = None
        self.merged = False

    @staticmethod
    def split(x):
        """-> (a, b, fork).  Pass `fork` to branch a's consumer (Conv(acc=fork) / UpsampleCatFn(.., fork)), then call
        ``b = GradFork.tap(b, fork)`` AFTER that consumer's forward: the autograd engine runs ready nodes in reverse creation
        order, so the tap -- created after branch a's node -- hands b's gradient over before branch a's backward looks for it."""
        if not (GRAD_FORK and torch.is_grad_enabled() and x.requires_grad):
            return x, x, None
        h = GradFork()
        a, b = _ForkFn.apply(x, h)
        return a, b, h

    @staticmethod
    def tap(b, fork):
        return b if fork is None else _TapFn.apply(b, fork)

    def take(self, shape, dtype):
        """the parked gradient of the other branch if this consumer can accumulate into it in place, else None"""
        g = self.gb
        if g is None or tuple(g.shape) != tuple(shape) or g.dtype != dtype or g.dim() != 4 or g.stride(3) != 1:
            return None
        ld = g.stride(2)
        if not (ld >= g.shape[3] and g.stride(1) == ld * g.shape[2] and g.stride(0) == ld * g.shape[2] * g.shape[1]):
            return None
        return g


class _ForkFn(Function):
    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        # exact aliases (view_as may re-derive the stride of a size-1 dimension, which the NHWC pixel-stride check relies on)
        return (x.as_strided(x.size(), x.stride(), x.storage_offset()), x.as_strided(x.size(), x.stride(), x.storage_offset()))

    @staticmethod
    def backward(ctx, ga, gb):
        h = ctx.holder
        merged = h.merged
        h.gb, h.merged = None, False
        if ga is None or gb is None:
            return (gb if ga is None else ga), None
        if merged:
            return ga, None                  # ga IS gb's buffer with both gradients in it
        return ga + gb, None


class _TapFn(Function):
    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.as_strided(x.size(), x.stride(), x.storage_offset())

    @staticmethod
    def backward(ctx, g):
        ctx.holder.gb = g
        ctx.holder.merged = False
        return g, None


def _wgrad(x, dy, cs, *more_slots):
    """Weight gradient of layer `cs` (grouped with same-shaped layers of this backward pass: ops.WgradQueue);
    the gradient-ready hook of every slot the launch covers fires once it has been issued."""
    hook = GRAD_READY_HOOK
    slots = (cs,) + more_slots

    def done():
        if hook is not None:
            for s in slots:
                hook(s)
    ops.WGRAD_QUEUE.submit(x, dy, cs.gw if not more_slots else _fused_gw(cs), cs.k, cs.stride, cs.pad,
                           on_done=done if hook is not None else None)


def _fused_gw(cs1):
    """gradient view of [cv1.w ; cv2.w] (C3StemFn): the two gradient slices are adjacent in the arena"""
    return cs1.gw.as_strided((2 * cs1.cout, 1, 1, cs1.cinp), (cs1.cinp, cs1.cinp, cs1.cinp, 1), cs1.gw.storage_offset())


class ConvBnActFn(Function):
    """z = act(BN_train(conv(x, w))) (+ residual)   -- reference Conv.forward (common.py:480-481) in
    train mode, plus the Bottleneck shortcut (common.py:544)."""

    @staticmethod
    def forward(ctx, x, residual, wparam, cs, bs, act, nbt, dst=None, bn_in=None, bn_out=None, acc=None):
        # wparam (the nn.Parameter) only ties the op into the autograd graph; its gradient is written
        # by the wgrad kernel directly into the flat arena, so backward returns None for it.
        # bn_in: BnBwdSums of the block that produced x, passed ONLY when this conv is x's sole consumer (the caller knows
        # the graph): this layer's dgrad then does that block's BatchNorm-backward reduce pass in its epilogue.
        # bn_out: a one-element list that receives this block's BnBwdSums for the (sole) consumer of z.
        # acc: the GradFork of x when x has a second consumer (see GradFork): this layer's dgrad adds into the parked gradient.
        ctx.w_needs_grad = wparam.requires_grad
        ctx.bn_in = bn_in if (bn_in is not None and cs.stride == 1 and residual is None) else None
        ctx.acc = acc
        if nbt is not None:
            nbt.add_(1)
        # dst = (buffer, channel offset): write the block output straight into its slice of a concat
        # buffer (JoinSlicesFn turns the filled buffer into the differentiable concat result)
        out = None if dst is None else dst[0][..., dst[1]:dst[1] + cs.coutp]
        y, z, scale, shift, mean, invstd = _bn_train_fwd(x, cs.w_lp, cs.stride, cs.pad, bs, act, residual=residual, out=out)
        ctx.cs, ctx.bs, ctx.act = cs, bs, act
        ctx.has_res = residual is not None
        ctx.x_needs_grad = x.requires_grad
        ctx.bn_mine = None
        if bn_out is not None and residual is None:
            ctx.bn_mine = _bn_sums(y, scale, shift, act, bs)
            bn_out.append(ctx.bn_mine)
        ctx.save_for_backward(x, y, scale, shift, mean, invstd)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, y, scale, shift, mean, invstd = ctx.saved_tensors
        cs, bs = ctx.cs, ctx.bs
        dz = _dense_or_slice(dz)
        dy = _bn_train_bwd(dz, y, bs, scale, shift, mean, invstd, ctx.act, sums=ctx.bn_mine)
        if ctx.w_needs_grad:
            _wgrad(x, dy, cs)
        dx = None
        if ctx.x_needs_grad:
            wT = cs.transposed()
            into = ctx.acc.take(x.shape, dy.dtype) if (ctx.acc is not None and ctx.bn_in is None) else None
            if into is not None:
                dx = ops.conv2d_dgrad(dy, wT, (x.shape[1], x.shape[2]), cs.stride, cs.pad, out=into, accumulate=True)
                ctx.acc.merged = True
            else:
                dx = ops.conv2d_dgrad(dy, wT, (x.shape[1], x.shape[2]), cs.stride, cs.pad, bn=_fuse_into(cs, ctx.bn_in, x))
        return dx, (dz if ctx.has_res else None), None, None, None, None, None, None, None, None, None


class BottleneckFn(Function):
    """z = x + cv2(cv1(x))   -- reference Bottleneck.forward with the shortcut (common.py:534-544), both Conv
    blocks in train mode.  One autograd node instead of two, so that the gradient of the shortcut is added in
    the epilogue of cv1's dgrad (dx = dgrad(dy1) + dz) instead of a separate accumulation pass by autograd."""

    @staticmethod
    def forward(ctx, x, w1, w2, cs1, bs1, cs2, bs2, act1, act2, nbt1, nbt2, dst=None, bn_in=None, bn_out=None):
        # bn_in / bn_out: see ConvBnActFn (x's sole consumer is this node: both the cv1 path and the shortcut are inside it)
        ctx.w_needs_grad = w1.requires_grad or w2.requires_grad
        ctx.bn_in = bn_in
        y1, h, *a1 = _bn_train_fwd(x, cs1.w_lp, cs1.stride, cs1.pad, bs1, act1)
        for nbt in (nbt1, nbt2):
            if nbt is not None:
                nbt.add_(1)
        out = None if dst is None else dst[0][..., dst[1]:dst[1] + cs2.coutp]
        y2, z, *a2 = _bn_train_fwd(h, cs2.w_lp, cs2.stride, cs2.pad, bs2, act2, residual=x, out=out)
        ctx.meta = (cs1, bs1, cs2, bs2, act1, act2)
        ctx.x_needs_grad = x.requires_grad
        ctx.bn_mine = None
        if bn_out is not None:
            ctx.bn_mine = _bn_sums(y2, a2[0], a2[1], act2, bs2)
            bn_out.append(ctx.bn_mine)
        ctx.save_for_backward(x, y1, h, y2, *a1, *a2)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, y1, h, y2, s1, b1, m1, i1, s2, b2, m2, i2 = ctx.saved_tensors
        cs1, bs1, cs2, bs2, act1, act2 = ctx.meta
        dz = _dense_or_slice(dz)
        dy2 = _bn_train_bwd(dz, y2, bs2, s2, b2, m2, i2, act2, sums=ctx.bn_mine)
        if ctx.w_needs_grad:
            _wgrad(h, dy2, cs2)
        # h = act(BN1(y1)) has exactly one consumer (cv2, inside this node): the reduce pass of BN1's backward rides
        # the epilogue of cv2's dgrad
        inner = _fuse_into(cs2, _bn_sums(y1, s1, b1, act1, bs1), h) if FUSE_BN_BWD else None
        dh = ops.conv2d_dgrad(dy2, cs2.transposed(), (h.shape[1], h.shape[2]), cs2.stride, cs2.pad, bn=inner)
        dy1 = _bn_train_bwd(dh, y1, bs1, s1, b1, m1, i1, act1, sums=inner)
        if ctx.w_needs_grad:
            _wgrad(x, dy1, cs1)
        dx = None
        if ctx.x_needs_grad:
            dx = ops.conv2d_dgrad(dy1, cs1.transposed(), (x.shape[1], x.shape[2]), cs1.stride, cs1.pad,
                                  residual=dz, bn=_fuse_into(cs1, ctx.bn_in, x))
        return (dx,) + (None,) * 13


def c3_stem_fusable(cs1, bs1, cs2, bs2):
    """cv1 and cv2 of a C3 can run as ONE GEMM when their weights / BN parameters / gradients / running
    statistics are adjacent in the flat arenas (they are: consecutive modules of the same shape) and no
    channel padding separates them."""
    es = cs1.w_lp.element_size()
    return (cs1.k == 1 and cs2.k == 1 and cs1.stride == 1 and cs2.stride == 1 and cs1.cinp == cs2.cinp
            and cs1.coutp == cs1.cout and cs2.coutp == cs2.cout and cs1.cout == cs2.cout
            and cs2.w_lp.data_ptr() == cs1.w_lp.data_ptr() + cs1.w_lp.numel() * es
            and cs2.gw.data_ptr() == cs1.gw.data_ptr() + cs1.gw.numel() * 4
            and bs2.aff_off == bs1.aff_off + bs1.c and bs1.c == cs1.cout and bs2.c == cs2.cout
            and bs1.eps == bs2.eps and bs1.momentum == bs2.momentum)


def _fused_vec(a, b):
    """[a | b] as one tensor: the two slices are adjacent in their arena"""
    return a.as_strided((a.numel() + b.numel(),), (1,), a.storage_offset())


class C3StemFn(Function):
    """(t, y2) = (cv1(x), cv2(x)) of a C3 block (common.py:589-591), both 1x1 Conv+BN+act on the SAME input, as
    one GEMM with the concatenated weights [cv1.w ; cv2.w]: x is read once, and in backward ONE dgrad over
    K = 2c_ produces the whole dx (autograd would otherwise add the two partial gradients in a separate
    pass) and one wgrad fills both weight gradients.  t goes to channels [0, c_) and y2 to [2c_, 3c_) of `buf`
    (N, H, W, 3c_): the bottleneck chain later writes its result to [c_, 2c_), so buf[..., c_:] is the
    concat [m(cv1(x)) | cv2(x)] that cv3 reads -- no copy."""

    @staticmethod
    def forward(ctx, x, w1, w2, cs1, bs1, cs2, bs2, act, nbt1, nbt2, buf, bn_out=None):
        # bn_out: receives the BnBwdSums of the cv1 half (t), whose sole consumer is the first bottleneck
        ctx.w_needs_grad = w1.requires_grad or w2.requires_grad
        c = cs1.cout
        wf = cs1.w_lp.as_strided((2 * c, 1, 1, cs1.cinp), (cs1.cinp, cs1.cinp, cs1.cinp, 1), cs1.w_lp.storage_offset())
        sh1 = sh2 = None
        if SHARDED_BN and bs1.sh_ld and bs2.sh_ld and ops.few_rows("fwd", x.dtype, x.shape[0], x.shape[1], x.shape[2], cs1.cinp, 2 * c, 1, 1, 0):
            sh1, sh2 = bs1.acquire_fwd(), bs2.acquire_fwd()
        for nbt in (nbt1, nbt2):
            if nbt is not None:
                nbt.add_(1)
        if sh1 is not None and sh2 is not None:
            # one GEMM adds the sums of both halves (adjacent channel ranges of the accumulator, c3_stem_fusable); each half folds its own
            y = ops.conv2d_fwd(x, wf, 1, 0, shards=sh1)
            N, H, W, _ = y.shape
            af = torch.empty((4, 2 * c), dtype=torch.float32, device=y.device)
            t = ops.bn_act_fwd_sharded(y[..., :c], sh1, N * H * W, bs1.gamma, bs1.beta, bs1.eps, bs1.momentum, bs1.rmean, bs1.rvar,
                                       act, out=buf[..., :c], aff=af[:, :c])[0]
            y2 = ops.bn_act_fwd_sharded(y[..., c:], sh2, N * H * W, bs2.gamma, bs2.beta, bs2.eps, bs2.momentum, bs2.rmean, bs2.rvar,
                                        act, out=buf[..., 2 * c:], aff=af[:, c:])[0]
            aff = (af[0], af[1], af[2], af[3])
        else:
            y, st = ops.conv2d_fwd(x, wf, 1, 0, want_stats=True)
            N, H, W, _ = y.shape
            aff = ops.bn_finalize(st, N * H * W, _fused_vec(bs1.gamma, bs2.gamma), _fused_vec(bs1.beta, bs2.beta), bs1.eps,
                                  bs1.momentum, _fused_vec(bs1.rmean, bs2.rmean), _fused_vec(bs1.rvar, bs2.rvar))
            t = ops.bn_act_fwd(y[..., :c], aff[0][:c], aff[1][:c], act, out=buf[..., :c])
            y2 = ops.bn_act_fwd(y[..., c:], aff[0][c:], aff[1][c:], act, out=buf[..., 2 * c:])
        ctx.meta = (cs1, bs1, cs2, bs2, act)
        ctx.x_needs_grad = x.requires_grad
        ctx.bn_mine = None
        if bn_out is not None:
            ctx.bn_mine = _bn_sums(y[..., :c], aff[0][:c], aff[1][:c], act, bs1)
            bn_out.append(ctx.bn_mine)
        ctx.save_for_backward(x, y, *aff)
        return t, y2

    @staticmethod
    def backward(ctx, dt, dy2):
        x, y, scale, shift, mean, invstd = ctx.saved_tensors
        cs1, bs1, cs2, bs2, act = ctx.meta
        c = cs1.cout
        N, H, W, _ = y.shape
        dy = torch.empty_like(y)
        halves = ((dt, slice(0, c), bs1), (dy2, slice(c, 2 * c), bs2))
        for g, sl, bs in halves:
            if g is None:
                dy[..., sl].zero_()
                continue
            g = _dense_or_slice(g)
            _bn_train_bwd(g, y[..., sl], bs, scale[sl], shift[sl], mean[sl], invstd[sl], act,
                          sums=ctx.bn_mine if sl.start == 0 else None, out=dy[..., sl])
        if ctx.w_needs_grad:
            _wgrad(x, dy, cs1, cs2)
        dx = None
        if ctx.x_needs_grad:
            wf = cs1.w_lp.as_strided((2 * c, 1, 1, cs1.cinp), (cs1.cinp, cs1.cinp, cs1.cinp, 1), cs1.w_lp.storage_offset())
            dx = ops.conv2d_dgrad(dy, ops.weight_transpose(wf), (x.shape[1], x.shape[2]), 1, 0)
        return (dx,) + (None,) * 11


class _GradDst:
    """Shared gradient buffer of one head output that was split along the batch (split_batch): each consumer
    (the fused loss) writes the gradient of ITS images straight into its range of `buf`; BatchSplitFn.backward
    then returns `buf` as the gradient of the whole tensor -- no zero-padded halves, no add."""

    def __init__(self, p):
        self.shape, self.stride, self.dtype, self.device = tuple(p.shape), tuple(p.stride()), p.dtype, p.device
        self.span = ops._flat_span(p)
        self.buf = None
        self.written = [False, False]

    def flat(self, half, n):
        """flat destination (elements) for images [0,n) (half 0) or [n,B) (half 1)"""
        if self.buf is None:
            self.buf = torch.empty(self.span, dtype=self.dtype, device=self.device)
        cut = n * self.stride[0]
        self.written[half] = True
        return self.buf[:cut] if half == 0 else self.buf[cut:]


class BatchSplitFn(Function):
    @staticmethod
    def forward(ctx, p, n, holder):
        ctx.n, ctx.holder = n, holder
        return p[:n], p[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        h, n = ctx.holder, ctx.n
        cut = n * h.stride[0]

        def is_slot(g, half):
            return (g is not None and h.buf is not None and h.written[half] and tuple(g.stride()) == h.stride
                    and g.data_ptr() == h.buf.data_ptr() + (cut * h.buf.element_size() if half else 0))
        if h.buf is None:
            h.buf = torch.empty(h.span, dtype=h.dtype, device=h.device)
        full = h.buf.as_strided(h.shape, h.stride)
        for half, g in ((0, ga), (1, gb)):
            if is_slot(g, half):
                continue
            dst = full[:n] if half == 0 else full[n:]
            if g is None:
                (h.buf[:cut] if half == 0 else h.buf[cut:]).zero_()
            else:
                (h.buf[:cut] if half == 0 else h.buf[cut:]).zero_()     # padding channels of the span
                dst.copy_(g)
        return full, None, None


def split_batch(p, n):
    """(p[:n], p[n:]) of a head output whose two halves feed two fused losses (reference
    SSODTrainer.split_predict_and_feature, ssod_trainer.py:570-585); the views carry `_et_grad_dst` so that the
    losses can deposit their gradients in place (see _GradDst)."""
    if not (torch.is_grad_enabled() and p.requires_grad) or n <= 0 or n >= p.shape[0]:
        return p[:n], p[n:]
    h = _GradDst(p)
    a, b = BatchSplitFn.apply(p, n, h)
    a._et_grad_dst = (h, 0, n)
    b._et_grad_dst = (h, 1, n)
    return a, b


class ConvBiasFn(Function):
    """y = act(conv(x, w) + bias): the Detect output convs (yolov5_head.py:30,55) and netD
    (yolo_ssod.py:224-238).  With ``head=(na, no)`` the result is returned as the (B, na, ny, nx, no)
    logits view of the NHWC GEMM output (yolov5_head.py:66 without the permute/contiguous copy)."""

    @staticmethod
    def forward(ctx, x, wparam, cs, act, head):
        ctx.w_needs_grad = wparam.requires_grad
        y = ops.conv2d_fwd(x, cs.w_lp, cs.stride, cs.pad, bias=cs.bias, act=act)
        ctx.cs, ctx.act, ctx.head = cs, act, head
        ctx.x_needs_grad = x.requires_grad
        ctx.save_for_backward(x, y)
        if head is None:
            return y
        return head_view(y, *head)

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        cs = ctx.cs
        if ctx.head is not None:
            dy = head_grad_to_nhwc(dy, y.shape, *ctx.head)
        dy = _dense_or_slice(dy)
        if ctx.act != ops.ACT_NONE:
            dy = ops.act_bwd(dy, y, ctx.act)
        if ctx.w_needs_grad:
            ops.conv2d_wgrad(x, dy, cs.gw, cs.k, cs.stride, cs.pad)
            if cs.gbias is not None:
                ops.colsum(dy, cs.gbias)
            if GRAD_READY_HOOK is not None:
                GRAD_READY_HOOK(cs)
        dx = None
        if ctx.x_needs_grad:
            wT = cs.transposed()
            dx = ops.conv2d_dgrad(dy, wT, (x.shape[1], x.shape[2]), cs.stride, cs.pad)
        return dx, None, None, None, None


def head_view(y, na, no):
    B, ny, nx, CP = y.shape
    return y.as_strided((B, na, ny, nx, no), (ny * nx * CP, no, nx * CP, CP, 1), y.storage_offset())


def head_grad_to_nhwc(g, yshape, na, no):
    """(B, na, ny, nx, no) gradient -> (B, ny, nx, CP).  The fused loss returns its gradient already in
    that memory layout (zero-copy); anything else is scattered into a zeroed buffer."""
    B, ny, nx, CP = yshape
    want = (ny * nx * CP, no, nx * CP, CP, 1)
    need = (g.storage_offset() + B * ny * nx * CP) * g.element_size()
    if tuple(g.stride()) == want and g.untyped_storage().nbytes() >= need:
        return g.as_strided((B, ny, nx, CP), (ny * nx * CP, nx * CP, CP, 1), g.storage_offset())
    buf = torch.zeros((B, ny, nx, CP), dtype=g.dtype, device=g.device)
    buf.as_strided((B, na, ny, nx, no), want).copy_(g)
    return buf


class JoinSlicesFn(Function):
    """torch.cat(parts, C) for parts that were PRODUCED IN PLACE as adjacent channel slices of ``buf``
    (C3.forward, common.py:590-591, without the copy).  Backward hands each producer its slice of the
    concat gradient as a view."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        off = 0
        for p in parts:
            assert p.data_ptr() == buf.data_ptr() + off * buf.element_size() and p.stride() == buf[..., :1].stride()
            off += p.shape[3]
        assert off == buf.shape[3]
        ctx.splits = [p.shape[3] for p in parts]
        return buf

    @staticmethod
    def backward(ctx, dcat):
        dcat = _dense_or_slice(dcat)
        return (None, *torch.split(dcat, ctx.splits, dim=3))


class SppfPoolFn(Function):
    """cat([x, m(x), m(m(x)), m(m(m(x)))], C)   -- SPPF.forward (common.py:702-708), m = MaxPool2d(5,1,2)."""

    @staticmethod
    def forward(ctx, x, holder=None):
        """holder = (cat,): x was PRODUCED IN PLACE as the first channel slice of the 4C-wide buffer (SPPF.forward passes
        cv1 that slot), so nothing is copied"""
        N, H, W, C = x.shape
        if holder is not None:
            cat = holder[0]
            assert cat.shape == (N, H, W, 4 * C) and x.data_ptr() == cat.data_ptr() and x.stride() == cat[..., :C].stride()
        else:
            cat = torch.empty((N, H, W, 4 * C), dtype=x.dtype, device=x.device)
            cat[..., :C].copy_(x)
        idx = []
        for i in range(3):
            _, ix = ops.maxpool5_fwd(cat[..., i * C:(i + 1) * C], out=cat[..., (i + 1) * C:(i + 2) * C])
            idx.append(ix)
        ctx.save_for_backward(*idx)
        ctx.C = C
        return cat

    @staticmethod
    def backward(ctx, dcat):
        i1, i2, i3 = ctx.saved_tensors
        C = ctx.C
        dcat = _dense_or_slice(dcat)
        d2 = ops.maxpool5_bwd(dcat[..., 3 * C:], i3, base=dcat[..., 2 * C:3 * C])
        d1 = ops.maxpool5_bwd(d2, i2, base=dcat[..., C:2 * C])
        return ops.maxpool5_bwd(d1, i1, base=dcat[..., :C]), None


class UpsampleCatFn(Function):
    """cat([upsample2x(a), b], C)   -- nn.Upsample + Concat in the neck (yolov5_neck.py:92-93, 97-98):
    the upsampled rows are written straight into the concat buffer."""

    @staticmethod
    def forward(ctx, a, b, acc=None, buf=None):
        """acc: the GradFork of `a` when `a` has a second consumer (the neck's lateral outputs also sit in a bottom-up concat).
        buf: the concat buffer `b` was produced in (YoloV5Neck.concat_slots), handed over explicitly by the caller"""
        N, H, W, Ca = a.shape
        Cb = b.shape[3]
        cat = in_place_concat_buffer(b, Ca, buf)
        if cat is None or cat.shape != (N, 2 * H, 2 * W, Ca + Cb):
            cat = torch.empty((N, 2 * H, 2 * W, Ca + Cb), dtype=a.dtype, device=a.device)
            cat[..., Ca:].copy_(b)
        ops.upsample2x_fwd(a, out=cat[..., :Ca])
        ctx.Ca = Ca
        ctx.acc = acc
        ctx.a_shape = tuple(a.shape)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        dcat = _dense_or_slice(dcat)
        da_src = dcat[..., :ctx.Ca]
        into = ctx.acc.take(ctx.a_shape, dcat.dtype) if ctx.acc is not None else None
        if into is not None:
            da = ops.upsample2x_bwd(da_src, out=into, accumulate=True)
            ctx.acc.merged = True
        else:
            da = ops.upsample2x_bwd(da_src)
        return da, dcat[..., ctx.Ca:], None, None


def in_place_concat_buffer(b, Ca, buf):
    """`b` was produced IN PLACE as channels [Ca, Ca + Cb) of the wider NHWC buffer `buf` (YoloV5Neck.concat_slots: the backbone's
    C3 / C4 blocks wrote P3 / P4 there): return `buf` if the geometry confirms it, else None (the caller then copies)."""
    if buf is None:
        return None
    es = buf.element_size()
    if (b.data_ptr() == buf.data_ptr() + Ca * es and b.stride() == buf[..., Ca:].stride() and b.shape[:3] == buf.shape[:3]
            and Ca + b.shape[3] == buf.shape[3]):
        return buf
    return None


class GradReverseFn(Function):  # reference models/detector/yolo_ssod.py:158-171
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ops.scale_inplace(g.contiguous().clone(), -1.0)


class DomainFocalFn(Function):
    """0.5 * mean over the pixels of all levels of -(1-p)^2 log p  (DomainLoss label 0 / TargetLoss
    label 1, models/loss/loss.py:376-421).  Inputs: the netD logits as (B,H,W,2) NHWC tensors."""

    @staticmethod
    def forward(ctx, label, *feats):
        total = sum(f.shape[0] * f.shape[1] * f.shape[2] for f in feats)
        acc = torch.zeros(1, dtype=torch.float32, device=feats[0].device)
        grads = [ops.domain_focal(f, label, 0.5 / total, acc, want_grad=True) for f in feats]
        ctx.save_for_backward(*grads)
        return ops.scale_cast(acc, torch.float32, scale=0.5 / total)

    @staticmethod
    def backward(ctx, gout):
        g = gout.reshape(1).float().contiguous()
        return (None, *[ops.scale_inplace(gr.clone(), 1.0, dev_scale=g) for gr in ctx.saved_tensors])


def _dense_or_slice(g):
    """Kernels need unit channel stride and pixels dense over one pixel stride; autograd may hand over
    expanded / broadcast gradients (e.g. zeros) -- materialise those."""
    if g.dim() == 4 and g.stride(3) == 1:
        ld = g.stride(2)
        if ld >= g.shape[3] and g.stride(1) == ld * g.shape[2] and g.stride(0) == ld * g.shape[2] * g.shape[1]:
            return g
    return g.contiguous()
