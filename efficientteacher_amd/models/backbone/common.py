"""Building blocks of the YOLOv5 graph (host-side mirror of reference models/backbone/common.py:
Conv :471, Bottleneck :534, C3 :566, SPPF :682, Concat :790, get_activation :28, autopad :50).

Same class names, constructor signatures, sub-module names (=> identical state_dict keys and
identical construction order => identical default initialisation for a given seed).  ``forward``
consumes / produces NHWC activations (N, H, W, C) and runs the gfx950 kernels: the ``nn.Conv2d`` /
``nn.BatchNorm2d`` children are parameter containers only, their own forward is never called.
"""
import torch
import torch.nn as nn

from ... import ops
from ... import autograd as _ag
from ...autograd import BottleneckFn, C3StemFn, ConvBnActFn, JoinSlicesFn, SppfPoolFn, c3_stem_fusable


_ACTIVATIONS = {"silu": (nn.SiLU, None), "relu": (lambda: nn.ReLU(inplace=True), "relu")}


def get_activation(act=True):
    """(module, act_name) as the reference's helper returns (models/backbone/common.py:28).  Only the activations the fused
    conv+BN+act kernels implement are constructible: SiLU (every shipped YOLOv5/v8 recipe), ReLU and none."""
    if isinstance(act, str):
        if act not in _ACTIVATIONS:
            raise AttributeError(f"activation '{act}' has no fused kernel (supported: {sorted(_ACTIVATIONS)})")
        make, name = _ACTIVATIONS[act]
        return make(), name
    if isinstance(act, nn.Module):
        return act, None
    return (nn.SiLU() if act is True else nn.Identity()), None


def autopad(k, p=None):
    """'same' padding for an odd kernel unless a padding is given"""
    if p is not None:
        return p
    return [x // 2 for x in k] if isinstance(k, (list, tuple)) else k // 2


def _act_code(m):
    if isinstance(m, nn.SiLU):
        return ops.ACT_SILU
    if isinstance(m, nn.ReLU):
        return ops.ACT_RELU
    if isinstance(m, nn.Identity):
        return ops.ACT_NONE
    raise NotImplementedError(f"activation {type(m).__name__} has no gfx950 kernel yet (SiLU / ReLU / Identity only)")


class Conv(nn.Module):
    # Standard convolution: act(bn(conv(x)))
    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):  # ch_in, ch_out, kernel, stride, padding, groups
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act, self.act_name = get_activation(act=act)

    def forward(self, x, residual=None, dst=None, bn_in=None, bn_out=None, acc=None):
        """dst = (buffer, channel offset): produce the output in place inside a wider NHWC buffer.
        bn_in / bn_out: BatchNorm-backward hand-over between a block and the SOLE consumer of its output
        (autograd.ConvBnActFn); only callers that know the graph pass them.
        acc: the autograd.GradFork of x when x has a second consumer: this layer's dgrad adds into that gradient in place."""
        cs = getattr(self.conv, "_et_slot", None)
        if cs is None:
            raise RuntimeError("model state is not on the device arenas yet: move the Model to a GPU "
                               "(model.to('cuda')) before calling it -- there is no CPU path")
        bs = self.bn._et_slot
        act = _act_code(self.act)
        if self.bn.training:
            nbt = None if self._et_flat().bulk_nbt else self.bn.num_batches_tracked   # bulk: bumped once per forward
            return ConvBnActFn.apply(x, residual, self.conv.weight, cs, bs, act, nbt, dst, bn_in, bn_out, acc)
        # eval (EMA teacher): BatchNorm is an affine of the running statistics, folded into the conv epilogue
        flat = self._et_flat()
        o = bs.aff_off
        out = None if dst is None else dst[0][..., dst[1]:dst[1] + cs.coutp]
        return ops.conv2d_fwd(x, cs.w_lp, cs.stride, cs.pad, scale=flat.eval_scale[o:o + bs.c],
                              bias=flat.eval_shift[o:o + bs.c], act=act, residual=residual, out=out)

    def forward_fuse(self, x):
        return self.forward(x)

    def _et_flat(self):
        return self.conv._et_flat_ref()


class Bottleneck(nn.Module):
    # Standard bottleneck
    def __init__(self, c1, c2, shortcut=True, g=1, k=(1, 3), e=0.5, act=True):
        super().__init__()
        c_ = int(c2 * e)  # hidden channels
        self.cv1 = Conv(c1, c_, k[0], 1, act=act)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g, act=act)
        self.add = shortcut and c1 == c2

    def forward(self, x, dst=None, bn_in=None, bn_out=None):
        """bn_in: the BatchNorm-backward hand-over of x's producer when this block is x's only consumer; bn_out: list
        that receives this block's own hand-over for the only consumer of its output (C3.forward knows both)."""
        c1, c2 = self.cv1, self.cv2
        if (self.add and c1.bn.training and c2.bn.training and torch.is_grad_enabled() and c1.conv.stride[0] == 1
                and getattr(c1.conv, "_et_slot", None) is not None):
            # shortcut + train mode: one fused autograd node (the shortcut gradient rides cv1's dgrad epilogue)
            bulk = c1._et_flat().bulk_nbt
            return BottleneckFn.apply(x, c1.conv.weight, c2.conv.weight, c1.conv._et_slot, c1.bn._et_slot,
                                      c2.conv._et_slot, c2.bn._et_slot, _act_code(c1.act), _act_code(c2.act),
                                      None if bulk else c1.bn.num_batches_tracked,
                                      None if bulk else c2.bn.num_batches_tracked, dst, bn_in, bn_out)
        if self.add or not (c1.bn.training and torch.is_grad_enabled() and _ag.FUSE_BN_BWD):
            return self.cv2(self.cv1(x), residual=x if self.add else None, dst=dst)
        mid = []                                  # cv1's output has one consumer: cv2
        h = self.cv1(x, bn_in=bn_in, bn_out=mid)
        return self.cv2(h, dst=dst, bn_in=mid[0] if mid else None, bn_out=bn_out)


class C3(nn.Module):
    # CSP Bottleneck with 3 convolutions
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, act=True):
        super().__init__()
        c_ = int(c2 * e)  # hidden channels
        if act == 'relu_silu':
            act, last_act = 'relu', 'silu'
        elif act == 'silu':
            act, last_act = 'silu', 'silu'
        elif act == 'relu_lrelu':
            act, last_act = 'relu', 'lrelu'
        elif act == 'relu_hswish':
            act, last_act = 'relu', 'hard_swish'
        else:
            last_act = act
        self.cv1 = Conv(c1, c_, 1, 1, act=act)
        self.cv2 = Conv(c1, c_, 1, 1, act=act)
        self.cv3 = Conv(2 * c_, c2, 1, act=last_act)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0, act=act) for _ in range(n)])

    def forward(self, x, dst=None):
        # cv3(cat(m(cv1(x)), cv2(x))): both halves are written in place into one buffer (no cat copy)
        # dst = (buffer, channel offset): cv3 writes the block's output straight into a slice of a wider buffer (the backbone's
        # P3 / P4 go into the neck's upsample-concat buffers this way: yolov5_neck.py:92-98 without the copy)
        N, H, W, _ = x.shape
        c_ = self.cv2.conv.out_channels
        c1s, c2s = getattr(self.cv1.conv, "_et_slot", None), getattr(self.cv2.conv, "_et_slot", None)
        last = len(self.m) - 1
        if (c1s is not None and self.cv1.bn.training and self.cv2.bn.training and torch.is_grad_enabled()
                and len(self.m) > 0 and c3_stem_fusable(c1s, self.cv1.bn._et_slot, c2s, self.cv2.bn._et_slot)):
            # train mode: cv1 | cv2 as one GEMM (C3StemFn); buf = [cv1(x) | m(cv1(x)) | cv2(x)]
            buf = torch.empty((N, H, W, 3 * c_), dtype=x.dtype, device=x.device)
            bulk = self.cv1._et_flat().bulk_nbt
            fuse = _ag.FUSE_BN_BWD
            hand = [] if fuse else None       # BatchNorm-backward hand-over along the chain t -> m[0] -> m[1] -> ...: every link
            t, y2 = C3StemFn.apply(x, self.cv1.conv.weight, self.cv2.conv.weight, c1s, self.cv1.bn._et_slot, c2s,   # has ONE consumer
                                   self.cv2.bn._et_slot, _act_code(self.cv1.act),
                                   None if bulk else self.cv1.bn.num_batches_tracked,
                                   None if bulk else self.cv2.bn.num_batches_tracked, buf, hand)
            for i, b in enumerate(self.m):
                nxt = [] if (fuse and i != last) else None
                t = b(t, dst=(buf, c_) if i == last else None, bn_in=hand[0] if hand else None, bn_out=nxt)
                hand = nxt
            return self.cv3(JoinSlicesFn.apply((buf[..., c_:],), t, y2), dst=dst)
        if (c1s is not None and not self.cv1.bn.training and not self.cv2.bn.training and not torch.is_grad_enabled() and len(self.m) > 0
                and c3_stem_fusable(c1s, self.cv1.bn._et_slot, c2s, self.cv2.bn._et_slot)
                and _act_code(self.cv1.act) == _act_code(self.cv2.act)):
            # eval without autograd (the EMA teacher): cv1 | cv2 as ONE GEMM with the folded BatchNorm affine in its epilogue -- x is
            # read once, one launch instead of two.  It writes [cv1(x) | cv2(x)] into the buffer cv3 reads; the LAST bottleneck then
            # writes m(cv1(x)) over the cv1 half (by then its only possible reader is that bottleneck's own shortcut add, which reads the
            # element it is about to overwrite in the same lane), so the buffer ends up as [m(cv1(x)) | cv2(x)]: no copy.
            flat = self.cv1._et_flat()
            o = self.cv1.bn._et_slot.aff_off
            buf = torch.empty((N, H, W, 2 * c_), dtype=x.dtype, device=x.device)
            wf = c1s.w_lp.as_strided((2 * c_, 1, 1, c1s.cinp), (c1s.cinp, c1s.cinp, c1s.cinp, 1), c1s.w_lp.storage_offset())
            ops.conv2d_fwd(x, wf, 1, 0, scale=flat.eval_scale[o:o + 2 * c_], bias=flat.eval_shift[o:o + 2 * c_],
                           act=_act_code(self.cv1.act), out=buf)
            t = buf[..., :c_]
            for i, b in enumerate(self.m):
                t = b(t, dst=(buf, 0) if i == last else None)
            return self.cv3(buf, dst=dst)
        buf = torch.empty((N, H, W, 2 * c_), dtype=x.dtype, device=x.device)
        y2 = self.cv2(x, dst=(buf, c_))
        train = self.cv1.bn.training and torch.is_grad_enabled() and _ag.FUSE_BN_BWD and len(self.m) > 0
        hand = [] if train else None
        t = self.cv1(x, bn_out=hand)
        for i, b in enumerate(self.m):
            nxt = [] if (train and i != last) else None
            t = b(t, dst=(buf, 0) if i == last else None, bn_in=hand[0] if hand else None, bn_out=nxt)
            hand = nxt
        if torch.is_grad_enabled() and (t.requires_grad or y2.requires_grad):
            cat = JoinSlicesFn.apply((buf,), t, y2)
        else:
            cat = buf
        return self.cv3(cat, dst=dst)


class C2f(nn.Module):
    # CSP Bottleneck with 2 convolutions (reference models/backbone/common.py:594-608): cv2(cat(split(cv1(x)), m_0, m_1, ...))
    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5, act=True):  # ch_in, ch_out, number, shortcut, groups, expansion
        super().__init__()
        self.c = int(c2 * e)  # hidden channels
        if self.c % 8:
            raise NotImplementedError("C2f hidden width must be a multiple of 8 channels (16-byte NHWC vectors)")
        self.cv1 = Conv(c1, 2 * self.c, 1, 1, act=act)
        self.cv2 = Conv((2 + n) * self.c, c2, 1, act=act)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=(3, 3), e=1.0, act=act) for _ in range(n))

    def forward(self, x):
        # every piece of the concat is produced in place in ONE buffer: cv1 fills [0, 2c), bottleneck i reads
        # [(1+i)c, (2+i)c) and writes [(2+i)c, (3+i)c); cv2 reads the whole buffer (no split / cat copies)
        N, H, W, _ = x.shape
        c, n = self.c, len(self.m)
        buf = torch.empty((N, H, W, (2 + n) * c), dtype=x.dtype, device=x.device)
        y = self.cv1(x, dst=(buf, 0))
        parts = [y]
        t = y[..., c:]
        for i, m in enumerate(self.m):
            t = m(t, dst=(buf, (2 + i) * c))
            parts.append(t)
        if torch.is_grad_enabled() and any(p.requires_grad for p in parts):
            return self.cv2(JoinSlicesFn.apply((buf,), *parts))
        return self.cv2(buf)


class SPPF(nn.Module):
    # Spatial Pyramid Pooling - Fast (SPPF) layer
    def __init__(self, c1, c2, k=5, act=True):
        super().__init__()
        if k != 5:
            raise NotImplementedError("the SPPF pooling kernel is specialised for k=5 (every shipped config)")
        c_ = c1 // 2
        if act == 'relu_silu':
            act, last_act = 'relu', 'silu'
        elif act == 'relu_lrelu':
            act, last_act = 'relu', 'lrelu'
        elif act == 'relu_hswish':
            act, last_act = 'relu', 'hard_swish'
        else:
            last_act = act
        self.cv1 = Conv(c1, c_, 1, 1, act=act)
        self.cv2 = Conv(c_ * 4, c2, 1, 1, act=last_act)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        # cv1 writes its output straight into the first quarter of the concat buffer the three poolings fill
        cs = getattr(self.cv1.conv, "_et_slot", None)
        if cs is None or cs.coutp != self.cv1.conv.out_channels:
            return self.cv2(SppfPoolFn.apply(self.cv1(x)))
        N, H, W, _ = x.shape
        cat = torch.empty((N, H, W, 4 * cs.coutp), dtype=x.dtype, device=x.device)
        return self.cv2(SppfPoolFn.apply(self.cv1(x, dst=(cat, 0)), (cat,)))


class Concat(nn.Module):
    # Concatenate a list of tensors along the channel dimension (NHWC: dim 3 <-> reference dim 1)
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, 3 if self.d == 1 else self.d)
