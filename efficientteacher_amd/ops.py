"""Thin tensor-level wrappers over the C ABI (include/et_hip.h).

Activations are NHWC tensors (N, H, W, C) -- possibly channel-slices of a wider buffer, i.e. the
last dim has stride 1 and the pixel stride is ``x.stride(2)``.  Weights are (Cout, KH, KW, Cin).
No arithmetic happens here: every function validates shapes, allocates outputs and launches.
"""
import torch

from . import _lib
from ._lib import ET_BF16, ET_F16, ET_F32

ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2


class KernelTimer:
    """Optional HIP-event timing of the MFMA conv launches on the launching stream (bench.py's roofline
    leg).  ``ops.TIMER = KernelTimer()`` switches it on; nothing is recorded otherwise."""

    def __init__(self):
        self.rows = []          # (tag, flops, launches, algorithmic bytes, start_event, end_event)
        self.shapes = []        # per row: (op, N, IH, IW, Cin, Cout, k, stride) or None

    def span(self, tag, flops, launches=1, nbytes=0, shape=None):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.rows.append((tag, flops, launches, nbytes, a, b))
        self.shapes.append(shape)
        return a, b

    def summary(self):
        agg = {}
        for tag, fl, n, nb, a, b in self.rows:
            ms = a.elapsed_time(b)
            t = agg.setdefault(tag, dict(ms=0.0, flops=0.0, launches=0, bytes=0.0))
            t["ms"] += ms; t["flops"] += fl; t["launches"] += n; t["bytes"] += nb
        return agg


TIMER = None
SCOPE = None          # "teacher" while the trainer issues the EMA teacher's forward (bench.py's per-family time budget)


class FamilyTimer:
    """HIP-event time budget of ONE step by kernel family (bench.py ``kernel_ms_by_family``): every launching entry point of
    the library is bracketed by two events on the stream it launches on (``_lib.CALL_TIMER``).  An event pair measures what
    the STREAM spends on the call: from the retirement of the stream's previous command to the end of the call's last
    kernel, so the pairs of one stream tile its busy time; what lies between them (torch's own kernels, launch gaps) is the
    remainder ``span - sum`` that bench.py reports as ``aten_and_gaps``."""

    FAMILY = {
        "et_conv2d_fwd": "gather_gemm", "et_conv2d_dgrad": "gather_gemm", "et_conv2d_dgrad_bn": "gather_gemm",
        "et_conv2d_wgrad": "wgrad", "et_conv2d_wgrad_grouped": "wgrad",
        "et_bn_finalize": "bn", "et_bn_act_fwd": "bn", "et_bn_act_fwd_sharded": "bn", "et_bn_act_bwd_sharded": "bn", "et_bn_act_bwd": "bn", "et_bn_act_bwd_from_partials": "bn", "et_act_bwd": "bn",
        "et_nms": "nms_loss_pl", "et_nms_ssod": "nms_loss_pl", "et_detect_decode": "nms_loss_pl", "et_pseudo_label_transform": "nms_loss_pl",
        "et_select_targets": "nms_loss_pl", "et_yolo_loss": "nms_loss_pl", "et_ota_assign": "nms_loss_pl", "et_score_log_append": "nms_loss_pl",
        "et_scale_cast": "nms_loss_pl", "et_domain_focal": "nms_loss_pl", "et_scale_inplace": "nms_loss_pl", "et_v8_decode": "nms_loss_pl",
        "et_tal_loss": "nms_loss_pl", "et_tal_assign": "nms_loss_pl", "et_colsum": "nms_loss_pl", "et_tal_pseudo_split": "nms_loss_pl", "et_tal_targets_pad": "nms_loss_pl",
        "et_tal_assigned_gt": "nms_loss_pl", "et_tal_merge_pseudo": "nms_loss_pl",
        "et_sgd_nesterov": "optimizer_ema", "et_sgd_nesterov_dev": "optimizer_ema", "et_adamw": "optimizer_ema", "et_ema_update": "optimizer_ema",
        "et_ema_update_dev": "optimizer_ema", "et_cast_f32_to_lp": "optimizer_ema", "et_scaler_check": "optimizer_ema", "et_scaler_update": "optimizer_ema", "et_weight_transpose": "optimizer_ema",
        "et_weight_transpose_all": "optimizer_ema", "et_bn_eval_affine": "optimizer_ema",
    }

    def __init__(self):
        self.rows = []          # (family, stream id, start event, end event)

    def begin(self, name):
        fam = self.FAMILY.get(name, "pool_upsample_pack")
        if fam == "gather_gemm":
            fam = "gather_gemm_teacher" if SCOPE == "teacher" else "gather_gemm_student"
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        return (fam, torch.cuda.current_stream().cuda_stream, a, b)

    def end(self, tok):
        tok[3].record()
        self.rows.append(tok)

    def summary(self):
        """{stream id: {family: ms}} -- call after a device synchronise"""
        out = {}
        for fam, sid, a, b in self.rows:
            d = out.setdefault(sid, {})
            d[fam] = d.get(fam, 0.0) + a.elapsed_time(b)
        return out


def kernel_name(op, dtype, N, IH, IW, Cin, Cout, k, stride, pad, parity_class=0, zero_page=True):
    """Name of the kernel instantiation the library launches for this conv problem (op: 'fwd' | 'dgrad' | 'wgrad', and
    'dgrad_full' / 'fwd_res' = the same with a residual / accumulate / BN-backward sums in the epilogue -- the persistent 1x1 kernel
    has separate instantiations for those; arguments of the FORWARD conv), as rocprofv3 prints it.  The selection lives in csrc/conv.hip only
    (et_conv2d_kernel_name); tests assert on it and bench.py tags its HIP-event timings with it."""
    import ctypes
    buf = ctypes.create_string_buffer(256)
    code = {"fwd": 0, "dgrad": 1, "wgrad": 2, "dgrad_full": 3, "fwd_res": 4}[op]
    dt = _ET_OF[dtype]
    _lib.check(_lib.load().et_conv2d_kernel_name(code, dt, N, IH, IW, Cin, Cout, k, k, stride, pad, int(bool(zero_page)),
                                                 parity_class, buf, 256), "et_conv2d_kernel_name")
    return buf.value.decode()


def stats_rows(op, dtype, N, IH, IW, Cin, Cout, k, stride, pad, zero_page=True):
    """rows of the partial-statistics buffer the selected kernel writes (op 'fwd': et_conv2d_fwd's stats_partial; 'fwd_res': the same
    for a call that also passes a residual; 'dgrad_bn': et_conv2d_dgrad_bn's bn_stats_partial; arguments of the FORWARD conv)"""
    dt = _ET_OF[dtype]
    rows = _lib.load().et_conv2d_stats_rows_for({"fwd": 0, "dgrad_bn": 1, "fwd_res": 2}[op], dt, N, IH, IW, Cin, Cout, k, k, stride, pad, int(bool(zero_page)))
    if rows <= 0:
        raise _lib.EtHipError(f"et_conv2d_stats_rows_for failed with code {rows}")
    return rows


def stats_adds(op, dtype, N, IH, IW, Cin, Cout, k, stride, pad, zero_page=True):
    """fp32 atomic additions per channel (all shards together) of the same call with a SHARDED accumulator: one per workgroup that
    covers the channel (et_conv2d_stats_adds_for)"""
    dt = _ET_OF[dtype]
    n = _lib.load().et_conv2d_stats_adds_for({"fwd": 0, "dgrad_bn": 1, "fwd_res": 2}[op], dt, N, IH, IW, Cin, Cout, k, k, stride, pad, int(bool(zero_page)))
    if n <= 0:
        raise _lib.EtHipError(f"et_conv2d_stats_adds_for failed with code {n}")
    return n


# Sharded statistics pay when the producer ADDS FEW times per channel (the persistent kernels: once per resident workgroup; the
# tiled kernels: once per row tile since r06, their wave rows meet in LDS first): every addition is one fp32 atomic per channel,
# adds / BN_SHARDS of them on one address.  Above the threshold the atomics serialise in the memory-side atomic units for longer than
# the finalize launch they replace (r05: every conv sharded = +5.3 ms of conv time against -1.6 ms of BatchNorm time per step; the
# threshold was 2048 partial ROWS then, with every wave adding).  12800 = the 128-row tiles of the 160 x 160 maps at 64 images: with one
# addition per workgroup EVERY layer of the YOLOv5l step is under it -- no finalize launch left in a 16-bit step (3200 -> 12800: -0.05 ms,
# inside the noise, profiles/r06_bn_shard_threshold_ab.txt; larger problems fall back to partial rows + finalize).
SHARD_MAX_ADDS = 12800
_FEW_ROWS = {}


def few_rows(op, dtype, N, IH, IW, Cin, Cout, k, stride, pad):
    key = (op, dtype, N, IH, IW, Cin, Cout, k, stride, pad)
    r = _FEW_ROWS.get(key)
    if r is None:
        r = _FEW_ROWS[key] = stats_adds(op, dtype, N, IH, IW, Cin, Cout, k, stride, pad)
    return r <= SHARD_MAX_ADDS


def few_reduce_rows(y):
    """the same question for the separate reduce pass of the BatchNorm backward (one addition per workgroup)"""
    N, H, W, C = y.shape
    return _lib.load().et_bn_reduce_rows(N * H * W, C, et_dtype(y)) <= SHARD_MAX_ADDS


def env_knobs():
    """The ET_* tuning knobs set in this process ("" = all defaults); recorded in bench.py's JSON line."""
    import ctypes
    buf = ctypes.create_string_buffer(1024)
    _lib.check(_lib.load().et_env_knobs(buf, 1024), "et_env_knobs")
    return buf.value.decode()


_ZERO_PAGES = {}


def zero_page(device):
    """256 zero bytes on `device`: the source of padding / tail lanes of the LDS-DMA conv kernels."""
    z = _ZERO_PAGES.get(device)
    if z is None:
        z = _ZERO_PAGES[device] = torch.zeros(256, dtype=torch.uint8, device=device)
    return z


_BN_TOTALS = {}


def bn_totals_scratch(device):
    """(2*2048 fp64 totals + 64 int32 tickets) zeros per (device, stream): the BN kernels need this scratch zero on
    entry and leave it zero on return (include/et_hip.h), so one buffer per stream replaces a memset launch and
    a finalize launch per layer."""
    key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    z = _BN_TOTALS.get(key)
    if z is None:
        z = _BN_TOTALS[key] = torch.zeros(2 * 2048 + 32, dtype=torch.float64, device=device)
    return z


# compute dtypes: float32 = parity mode (exact-f32 MFMA), bfloat16 = performance mode, float16 = the reference's own reduced precision
# (torch.cuda.amp autocast + GradScaler, trainer/trainer.py:248,348,399-400: same MFMA rate as bf16, needs the loss scaler below)
_ET_OF = {torch.float32: ET_F32, torch.bfloat16: ET_BF16, torch.float16: ET_F16}
LP_DTYPES = (torch.bfloat16, torch.float16)


def et_dtype(t):
    try:
        return _ET_OF[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (float32 = parity mode, bfloat16 = performance mode, float16 = the "
                        "reference's AMP recipe)") from None


def _nhwc(x):
    """(N,H,W,C) with unit channel stride and dense N,H,W over a pixel stride -> pixel stride."""
    assert x.dim() == 4 and x.stride(3) == 1, "NHWC tensor with contiguous channels expected"
    ld = x.stride(2)
    assert x.stride(1) == ld * x.shape[2] and x.stride(0) == ld * x.shape[2] * x.shape[1], \
        "pixels must be dense over the pixel stride"
    return ld


def conv_out_hw(ih, iw, k, s, p):
    return (ih + 2 * p - k) // s + 1, (iw + 2 * p - k) // s + 1


def conv2d_fwd(x, w, stride, pad, *, scale=None, bias=None, act=ACT_NONE, residual=None, out=None, want_stats=False, shards=None):
    """y = act(conv(x, w) * scale + bias) + residual ; optional BN partial statistics (rows, 2, Cout) -- or, with shards = (tensor
    whose first element is this layer's channel 0 of a zeroed [BN_SHARDS][2][ld] accumulator, ld), ADDED into that accumulator."""
    N, IH, IW, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2 and w.is_contiguous() and w.dtype == x.dtype
    OH, OW = conv_out_hw(IH, IW, KH, stride, pad)
    lib = _lib.load()
    if out is None:
        out = torch.empty((N, OH, OW, Cout), dtype=x.dtype, device=x.device)
    assert out.shape == (N, OH, OW, Cout)
    stats, stats_ld = None, 0
    if shards is not None:
        assert not want_stats
        stats, stats_ld = shards
    if want_stats:
        rows = stats_rows("fwd_res" if residual is not None else "fwd", x.dtype, N, IH, IW, Cin, Cout, KH, stride, pad)
        stats = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
    ldr = _nhwc(residual) if residual is not None else 0
    ev = TIMER.span(kernel_name("fwd_res" if residual is not None else "fwd", x.dtype, N, IH, IW, Cin, Cout, KH, stride, pad), 2.0 * N * OH * OW * Cout * Cin * KH * KW,
                    nbytes=(x.numel() + N * OH * OW * Cout + w.numel()) * x.element_size(),
                    shape=("fwd", N, IH, IW, Cin, Cout, KH, stride)) if TIMER else None
    if ev:
        ev[0].record()
    _lib.check(lib.et_conv2d_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(out), et_dtype(x), N, IH, IW, Cin, _nhwc(x),
                                 Cout, KH, KW, stride, pad, _nhwc(out), _lib.ptr(scale), _lib.ptr(bias), act,
                                 _lib.ptr(residual), ldr, _lib.ptr(stats), stats_ld, _lib.ptr(zero_page(x.device)),
                                 _lib.stream(x)), "et_conv2d_fwd")
    if ev:
        ev[1].record()
    return (out, stats) if want_stats else out


class WgradQueue:
    """Deferred, grouped weight-gradient launches (et_conv2d_wgrad_grouped) on a SIDE STREAM.

    Inside a backward pass, layers of identical geometry (the bottleneck stacks of a YOLOv5 stage) are collected
    and launched together: the K-split that fills the chip is shared by the group, so every dW address gets
    group-size times fewer fp32 atomics.  A group is launched when it reaches ``group`` items and at the end of
    the backward pass (autograd engine callback); ``on_done`` callbacks (gradient-ready hooks of the data-parallel
    wrapper) run right after the launch that covers their layer.  Groups of 8 (8 / 4 / 2 / 1 measured 54.73 / 54.84 / 55.44 /
    56.75 ms on the step, profiles/r03_wgrad_ident_and_group_ab.txt; ``group = 1`` launches every layer at once).

    Nothing in backward consumes a weight gradient, so on a GPU the grouped launches CAN go to a second HIP stream
    (ET_WGRAD_STREAM=1): the MFMA-bound wgrad workgroups then run beside the critical path (dgrad -> BatchNorm backward -> dgrad ...)
    instead of inside it.  Measured on the YOLOv5l SSOD step, alternating runs on one box (profiles/r03_wgrad_stream_ab.txt):
    58.13 / 58.10 ms against 58.77 / 58.75 ms -- about 1 %.  The GPU is close to work-conserving here: with the 12.6 ms of wgrad
    kernels off the main stream its own kernels stretch by ~6 ms (BatchNorm passes 14.1 -> 17.2 ms, gather-GEMMs 26.0 -> 29.2 ms),
    which is why the gain is 0.6 ms and not 12 -- and why the PER-LAUNCH figures get worse while the step gets faster (the
    dominant gather-GEMM's HIP-event duration 132 -> 149 us, bench.py `roofline.frac` 0.23 -> 0.21).  r03-r05 therefore kept the
    launches on the launching stream.  r06: the DEFAULT is the side stream (ET_WGRAD_STREAM=0 restores the launching stream) -- re-measured
    at the current kernels, 50.43 -> 50.10 ms over 100-step runs and 49.81 -> 49.55 ms over 20-step runs, alternating on one box
    (profiles/r06_wgrad_side_stream_ab.txt): the step runs at the chip's power limit (profiles/r06_power_limit.txt), where what a change is
    worth is decided on the step, not per launch; bench.py reports the dominant kernel's uncontended duration beside the in-region one
    (`roofline.same_kernel_nothing_co_resident`: one extra step with the teacher and the weight gradients on the main stream).  Ordering: the side stream waits for the
    launching stream at every group launch (dy and x are complete), the launching stream joins the side stream at the end of
    backward (before the optimizer / the final all-reduces); gradient-ready hooks run with the side stream current, so an RCCL
    all-reduce they start is ordered behind the wgrads it covers."""

    def __init__(self):
        import os
        self.group = 8
        self.stale = 20              # a group nobody added to for this many submissions is launched (its stage of
        self.pending = {}            # the network is over): keeps the gradient all-reduce overlapped with backward
        self.last = {}
        self.tick = 0
        self._cb_armed = False
        self.use_side = int(os.environ.get("ET_WGRAD_STREAM", "1") or 0)     # 0 launching stream, 1 every group (default), 2 only the 1x1 layers, 3 only k > 1
        self._side = {}              # device -> side stream
        self._dirty = set()          # devices whose side stream holds work the launching stream has not joined yet

    def side_stream(self, dev):
        s = self._side.get(dev)
        if s is None:
            s = self._side[dev] = torch.cuda.Stream(device=dev)
        return s

    def reset(self):
        """Start of a training forward: nothing of an earlier (possibly failed) backward may linger -- pending
        entries of a pass that raised would otherwise be flushed into a later step's gradient arena, and a stale
        armed flag would keep later passes from ever arming their end-of-backward flush."""
        self.pending.clear()
        self.last.clear()
        self._cb_armed = False
        self.join()

    def submit(self, x, dy, dw, ksize, stride, pad, on_done=None):
        if self.group <= 1 or x.dtype not in LP_DTYPES:
            conv2d_wgrad(x, dy, dw, ksize, stride, pad)
            if on_done is not None:
                on_done()
            return
        if not self._cb_armed:       # one end-of-backward callback per pass: flush the partial groups, join the side stream
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
                self._cb_armed = True
            except RuntimeError:     # not inside a backward pass: nothing will call back -- launch now, in order
                conv2d_wgrad(x, dy, dw, ksize, stride, pad)
                if on_done is not None:
                    on_done()
                return
        key = (tuple(x.shape), tuple(dy.shape), ksize, stride, pad, x.device)
        self.tick += 1
        for k in [k for k, t in self.last.items() if k != key and self.tick - t > self.stale]:
            self._flush_key(k)
        lst = self.pending.setdefault(key, [])
        lst.append((x, dy, dw, on_done))
        self.last[key] = self.tick
        if len(lst) >= self.group:
            self._flush_key(key)

    def _flush_key(self, key):
        lst = self.pending.pop(key, None)
        self.last.pop(key, None)
        if not lst:
            return
        xs, dys, ksize, stride, pad, dev = key
        items = [(a, b, c) for a, b, c, _ in lst]
        if dev.type == "cuda" and (self.use_side == 1 or (self.use_side == 2 and ksize == 1) or (self.use_side == 3 and ksize > 1)):
            main, side = torch.cuda.current_stream(dev), self.side_stream(dev)
            side.wait_stream(main)               # every x / dy of the group is complete on the launching stream
            with torch.cuda.stream(side):
                conv2d_wgrad_grouped(items, ksize, stride, pad)
                for a, b, _ in items:            # the caching allocator must not hand these blocks out before the side
                    a.record_stream(side)        # stream has read them (autograd frees them when the node returns)
                    b.record_stream(side)
                for _, _, _, cb in lst:
                    if cb is not None:
                        cb()
            self._dirty.add(dev)
            return
        conv2d_wgrad_grouped(items, ksize, stride, pad)
        for _, _, _, cb in lst:
            if cb is not None:
                cb()

    def join(self):
        """the launching stream waits for everything the side stream(s) hold (end of backward / before the optimizer)"""
        for dev in self._dirty:
            torch.cuda.current_stream(dev).wait_stream(self._side[dev])
        self._dirty.clear()

    def flush(self):
        self._cb_armed = False
        for key in list(self.pending.keys()):
            self._flush_key(key)
        self.join()


WGRAD_QUEUE = WgradQueue()


def conv2d_wgrad_grouped(items, ksize, stride, pad):
    """items: [(x, dy, dw)] of identical shapes; dw (Cout, KH, KW, Cin) fp32 += wgrad(x, dy) for each, one launch."""
    x0, dy0, dw0 = items[0]
    N, IH, IW, Cin = x0.shape
    _, OH, OW, Cout = dy0.shape
    arr = (_lib.WgradItem * len(items))()
    for i, (x, dy, dw) in enumerate(items):
        assert x.shape == x0.shape and dy.shape == dy0.shape and x.dtype == dy.dtype == x0.dtype
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.shape == (Cout, ksize, ksize, Cin)
        arr[i].x, arr[i].dy, arr[i].dw = _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw)
        arr[i].ldx, arr[i].ldy = _nhwc(x), _nhwc(dy)
    # algorithmic bytes of a weight gradient: x and dy read once, dw (fp32) read-modify-written once -- per item of the group
    ev = TIMER.span(kernel_name("wgrad", x0.dtype, N, IH, IW, Cin, Cout, ksize, stride, pad),
                    2.0 * len(items) * N * OH * OW * Cout * Cin * ksize * ksize, 1,
                    nbytes=len(items) * ((N * IH * IW * Cin + N * OH * OW * Cout) * x0.element_size() + 8 * Cout * Cin * ksize * ksize),
                    shape=("wgrad", N, IH, IW, Cin, Cout, ksize, stride)) if TIMER else None
    if ev:
        ev[0].record()
    _lib.check(_lib.load().et_conv2d_wgrad_grouped(arr, len(items), et_dtype(x0), N, IH, IW, Cin, Cout, ksize, ksize,
                                                   stride, pad, _lib.ptr(zero_page(x0.device)), _lib.stream(x0)),
               "et_conv2d_wgrad_grouped")
    if ev:
        ev[1].record()


def weight_transpose(w):
    """(Cout, KH, KW, Cin) -> (Cin, KH, KW, Cout), the dgrad operand."""
    Cout, KH, KW, Cin = w.shape
    wt = torch.empty((Cin, KH, KW, Cout), dtype=w.dtype, device=w.device)
    _lib.check(_lib.load().et_weight_transpose(_lib.ptr(w), _lib.ptr(wt), et_dtype(w), Cout, KH * KW, Cin,
                                               _lib.stream(w)), "et_weight_transpose")
    return wt


def weight_transpose_all(w_arena, wT_arena, table, total):
    """Transpose every layer of the flat weight arena (table: (n, 4) int32 rows {offset, Cout, taps, Cin})."""
    _lib.check(_lib.load().et_weight_transpose_all(_lib.ptr(w_arena), _lib.ptr(wT_arena), et_dtype(w_arena),
                                                   _lib.ptr(table), table.shape[0], int(total), _lib.stream(w_arena)),
               "et_weight_transpose_all")


class BnBwdSums:
    """What the dgrad of a CONSUMER layer needs to run the reduce pass of its PRODUCER's BatchNorm backward in its own
    epilogue (et_conv2d_dgrad_bn): the producer's raw conv output y, folded affine and activation -- and, after that
    dgrad has run, the partial sums it left plus the identity of the tensor they belong to."""
    __slots__ = ("y", "scale", "shift", "act", "partial", "dz_ptr", "slot")

    def __init__(self, y, scale, shift, act, slot=None):
        # slot: the producer's flat_state.BnSlot when its backward sums go to the sharded accumulator (no partial rows, no finalize
        # launch): the dgrad that carries the sums acquires the shards, `partial` is then the (tensor, ld) pair it added into
        self.y, self.scale, self.shift, self.act, self.slot = y, scale, shift, act, slot
        self.partial, self.dz_ptr = None, None

    def take(self, dz):
        """the partial sums, if `dz` is exactly the tensor they were computed on (else None: fall back to the reduce pass)"""
        p = self.partial if (self.partial is not None and dz.data_ptr() == self.dz_ptr) else None
        self.partial, self.dz_ptr = None, None
        return p


def conv2d_dgrad(dy, wT, in_hw, stride, pad, *, out=None, accumulate=False, residual=None, bn=None):
    """dx (N, IH, IW, Cin) from dy (N, OH, OW, Cout) and wT (Cin, KH, KW, Cout).  bn (a BnBwdSums): dx is the activation
    gradient of a Conv block whose BatchNorm-backward reduce pass runs in this launch's epilogue (stride 1 only)."""
    N, OH, OW, Cout = dy.shape
    Cin, KH, KW, Cout2 = wT.shape
    assert Cout == Cout2 and wT.dtype == dy.dtype
    IH, IW = in_hw
    assert conv_out_hw(IH, IW, KH, stride, pad) == (OH, OW)
    if out is None:
        assert not accumulate
        out = torch.empty((N, IH, IW, Cin), dtype=dy.dtype, device=dy.device)
    if bn is not None and stride == 1 and not accumulate and bn.y.shape == out.shape and bn.y.dtype == out.dtype:
        lib = _lib.load()
        if bn.slot is not None and few_rows("dgrad_bn", dy.dtype, N, IH, IW, Cin, Cout, KH, stride, pad):
            keep = bn.slot.acquire_bwd()
            part, part_ld = keep
        else:
            rows = stats_rows("dgrad_bn", dy.dtype, N, IH, IW, Cin, Cout, KH, stride, pad)
            keep = part = torch.empty((rows, 2, Cin), dtype=torch.float32, device=dy.device)
            part_ld = 0
        ev = TIMER.span(kernel_name("dgrad_full", dy.dtype, N, IH, IW, Cin, Cout, KH, stride, pad), 2.0 * N * OH * OW * Cout * Cin * KH * KW,
                        nbytes=(dy.numel() + 2 * N * IH * IW * Cin + wT.numel()) * dy.element_size(),
                        shape=("dgrad", N, IH, IW, Cin, Cout, KH, stride)) if TIMER else None
        if ev:
            ev[0].record()
        _lib.check(lib.et_conv2d_dgrad_bn(_lib.ptr(dy), _lib.ptr(wT), _lib.ptr(out), et_dtype(dy), N, IH, IW, Cin, _nhwc(out), Cout,
                                          KH, KW, pad, _nhwc(dy), _lib.ptr(residual), _nhwc(residual) if residual is not None else 0,
                                          _lib.ptr(bn.y), _nhwc(bn.y), _lib.ptr(bn.scale), _lib.ptr(bn.shift), bn.act,
                                          _lib.ptr(part), part_ld, _lib.ptr(zero_page(dy.device)), _lib.stream(dy)), "et_conv2d_dgrad_bn")
        if ev:
            ev[1].record()
        bn.partial, bn.dz_ptr = keep, out.data_ptr()
        return out
    tag = (kernel_name("dgrad_full" if (accumulate or residual is not None) else "dgrad", dy.dtype, N, IH, IW, Cin, Cout, KH, stride, pad) if stride == 1 else
           "conv_gemm (stride-2 dgrad parity classes)") if TIMER else None
    ev = TIMER.span(tag, 2.0 * N * OH * OW * Cout * Cin * KH * KW,
                    stride * stride, nbytes=(dy.numel() + N * IH * IW * Cin + wT.numel()) * dy.element_size(),
                    shape=("dgrad", N, IH, IW, Cin, Cout, KH, stride)) if TIMER else None
    if ev:
        ev[0].record()
    _lib.check(_lib.load().et_conv2d_dgrad(_lib.ptr(dy), _lib.ptr(wT), _lib.ptr(out), et_dtype(dy), N, IH, IW, Cin,
                                           _nhwc(out), Cout, KH, KW, stride, pad, _nhwc(dy), int(accumulate),
                                           _lib.ptr(residual), _nhwc(residual) if residual is not None else 0,
                                           _lib.ptr(zero_page(dy.device)), _lib.stream(dy)), "et_conv2d_dgrad")
    if ev:
        ev[1].record()
    return out


def conv2d_wgrad(x, dy, dw, ksize, stride, pad):
    """dw (Cout, KH, KW, Cin) fp32 += wgrad(x, dy)   (accumulates: dw is the gradient arena slice)."""
    N, IH, IW, Cin = x.shape
    _, OH, OW, Cout = dy.shape
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.shape == (Cout, ksize, ksize, Cin)
    assert x.dtype == dy.dtype
    ev = TIMER.span(kernel_name("wgrad", x.dtype, N, IH, IW, Cin, Cout, ksize, stride, pad), 2.0 * N * OH * OW * Cout * Cin * ksize * ksize) if TIMER else None
    if ev:
        ev[0].record()
    _lib.check(_lib.load().et_conv2d_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), et_dtype(x), N, IH, IW, Cin,
                                           _nhwc(x), Cout, ksize, ksize, stride, pad, _nhwc(dy),
                                           _lib.ptr(zero_page(x.device)), _lib.stream(x)), "et_conv2d_wgrad")
    if ev:
        ev[1].record()
    return dw


def colsum(x2d_like, out):
    """out[c] += sum over pixels of an NHWC tensor."""
    N, H, W, C = x2d_like.shape
    _lib.check(_lib.load().et_colsum(_lib.ptr(x2d_like), et_dtype(x2d_like), N * H * W, C, _nhwc(x2d_like),
                                     _lib.ptr(out), _lib.stream(out)), "et_colsum")
    return out


# ---- BatchNorm + activation -----------------------------------------------------------------------
def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean=None, running_var=None):
    rows, _, C = stats.shape
    dev = stats.device
    aff = torch.empty((4, C), dtype=torch.float32, device=dev)
    scale, shift, mean, invstd = aff[0], aff[1], aff[2], aff[3]
    assert C <= 2048
    ws = bn_totals_scratch(dev)
    _lib.check(_lib.load().et_bn_finalize(_lib.ptr(stats), rows, C, float(count), _lib.ptr(gamma), _lib.ptr(beta),
                                          eps, momentum, _lib.ptr(running_mean), _lib.ptr(running_var),
                                          _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd),
                                          _lib.ptr(ws), _lib.stream(stats)), "et_bn_finalize")
    return scale, shift, mean, invstd


def bn_eval_affine(gamma, beta, running_mean, running_var, eps):
    C = gamma.numel()
    scale = torch.empty(C, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    _lib.check(_lib.load().et_bn_eval_affine(C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                             _lib.ptr(running_var), eps, _lib.ptr(scale), _lib.ptr(shift),
                                             _lib.stream(gamma)), "et_bn_eval_affine")
    return scale, shift


def bn_act_fwd(y, scale, shift, act, residual=None, out=None):
    N, H, W, C = y.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=y.dtype, device=y.device)
    ldr = _nhwc(residual) if residual is not None else 0
    _lib.check(_lib.load().et_bn_act_fwd(_lib.ptr(y), _nhwc(y), _lib.ptr(out), _nhwc(out), _lib.ptr(residual), ldr,
                                         et_dtype(y), N * H * W, C, _lib.ptr(scale), _lib.ptr(shift), act,
                                         _lib.stream(y)), "et_bn_act_fwd")
    return out


def bn_act_fwd_sharded(y, shards, count, gamma, beta, eps, momentum, running_mean, running_var, act, residual=None, out=None, aff=None):
    """bn_finalize + bn_act_fwd in ONE launch on the sharded sums of conv2d_fwd(shards=...): (z, scale, shift, mean, invstd);
    aff: a (4, C) fp32 view that receives the four per-channel vectors (else allocated)"""
    N, H, W, C = y.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=y.dtype, device=y.device)
    if aff is None:
        aff = torch.empty((4, C), dtype=torch.float32, device=y.device)
    ldr = _nhwc(residual) if residual is not None else 0
    sh, ld = shards
    a0 = _lib.ptr(aff)                                  # rows of the (4, C) block by address: four view tensors per call were host time
    rs = aff.stride(0) * 4
    assert aff.stride(1) == 1 and aff.dtype == torch.float32
    _lib.check(_lib.load().et_bn_act_fwd_sharded(_lib.ptr(y), _nhwc(y), _lib.ptr(out), _nhwc(out), _lib.ptr(residual), ldr, et_dtype(y),
                                                 N * H * W, C, _lib.ptr(sh), ld, float(count), _lib.ptr(gamma), _lib.ptr(beta), eps,
                                                 momentum, _lib.ptr(running_mean), _lib.ptr(running_var), a0, a0 + rs, a0 + 2 * rs,
                                                 a0 + 3 * rs, act, _lib.stream(y)),
               "et_bn_act_fwd_sharded")
    sc, shf, mean, invstd = aff.unbind(0)
    return out, sc, shf, mean, invstd


def bn_act_bwd(dz, y, gamma, scale, shift, mean, invstd, act, dgamma, dbeta, out=None, partial=None, shards=None):
    """partial: (rows, 2, C) sums left by the dgrad that produced dz (BnBwdSums.take) -- the reduce pass is skipped.
    shards = (tensor, ld): the sharded form, no finalize launch: `partial` (the same pair, from the dgrad) means the sums are there
    already, else the reduce pass adds them into the (zeroed) shards first."""
    N, H, W, C = y.shape
    lib = _lib.load()
    if out is None:
        out = torch.empty((N, H, W, C), dtype=y.dtype, device=y.device)
    if shards is not None:
        sh, ld = shards
        _lib.check(lib.et_bn_act_bwd_sharded(_lib.ptr(dz), _nhwc(dz), _lib.ptr(y), _nhwc(y), _lib.ptr(out), _nhwc(out), et_dtype(y),
                                             N * H * W, C, _lib.ptr(gamma), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean),
                                             _lib.ptr(invstd), act, _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(sh), ld,
                                             0 if partial is not None else 1, _lib.stream(y)), "et_bn_act_bwd_sharded")
        return out
    if partial is not None:
        ws = torch.empty(3 * C, dtype=torch.float32, device=y.device)
        _lib.check(lib.et_bn_act_bwd_from_partials(_lib.ptr(dz), _nhwc(dz), _lib.ptr(y), _nhwc(y), _lib.ptr(out), _nhwc(out),
                                                   et_dtype(y), N * H * W, C, _lib.ptr(gamma), _lib.ptr(scale), _lib.ptr(shift),
                                                   _lib.ptr(mean), _lib.ptr(invstd), act, _lib.ptr(dgamma), _lib.ptr(dbeta),
                                                   _lib.ptr(bn_totals_scratch(y.device)), _lib.ptr(partial), partial.shape[0],
                                                   _lib.ptr(ws), _lib.stream(y)), "et_bn_act_bwd_from_partials")
        return out
    rows = lib.et_bn_reduce_rows(N * H * W, C, et_dtype(y))
    nws = rows * 2 * C + 3 * C
    ws = torch.empty(nws, dtype=torch.float32, device=y.device)
    _lib.check(lib.et_bn_act_bwd(_lib.ptr(dz), _nhwc(dz), _lib.ptr(y), _nhwc(y), _lib.ptr(out), _nhwc(out),
                                 et_dtype(y), N * H * W, C, _lib.ptr(gamma), _lib.ptr(scale), _lib.ptr(shift),
                                 _lib.ptr(mean), _lib.ptr(invstd), act, _lib.ptr(dgamma), _lib.ptr(dbeta),
                                 _lib.ptr(bn_totals_scratch(y.device)), _lib.ptr(ws), nws, _lib.stream(y)),
               "et_bn_act_bwd")
    return out


def act_bwd(dz, y, act, out=None):
    N, H, W, C = y.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=y.dtype, device=y.device)
    _lib.check(_lib.load().et_act_bwd(_lib.ptr(dz), _nhwc(dz), _lib.ptr(y), _nhwc(y), _lib.ptr(out), _nhwc(out),
                                      et_dtype(y), N * H * W, C, act, _lib.stream(y)), "et_act_bwd")
    return out


# ---- spatial ----------------------------------------------------------------------------------------
def pack_input(x_nchw, dtype, norm_scale=255.0, out=None):
    """(B,3,H,W) NCHW -> (B,H,W,8) NHWC of `dtype`, channels zero padded.  fp32 input: values as they are (already
    normalised by the caller); uint8 input (the loaders' batches): (float)x / norm_scale in the same pass."""
    if isinstance(x_nchw, (list, tuple)):
        # several batches of one shape (the labelled and the unlabelled images of an SSOD step): packed into consecutive
        # ranges of ONE buffer -- the reference's torch.cat of the two batches (ssod_trainer.py:623) without the copy
        parts = list(x_nchw)
        _, C, H, W = parts[0].shape
        assert all(p.shape[1:] == parts[0].shape[1:] and p.device == parts[0].device for p in parts)
        y = torch.empty((sum(p.shape[0] for p in parts), H, W, 8), dtype=dtype, device=parts[0].device)
        b0 = 0
        for p in parts:
            pack_input(p, dtype, norm_scale, out=y[b0:b0 + p.shape[0]])
            b0 += p.shape[0]
        return y
    x = x_nchw.contiguous()
    B, C, H, W = x.shape
    y = out if out is not None else torch.empty((B, H, W, 8), dtype=dtype, device=x.device)
    assert y.shape == (B, H, W, 8) and y.is_contiguous() and y.dtype == dtype
    if x.dtype == torch.uint8:
        _lib.check(_lib.load().et_pack_input_u8(_lib.ptr(x), _lib.ptr(y), et_dtype(y), B, C, H, W, float(norm_scale), _lib.stream(x)),
                   "et_pack_input_u8")
        return y
    if x.dtype != torch.float32:
        x = x.float()
    _lib.check(_lib.load().et_pack_input(_lib.ptr(x), _lib.ptr(y), et_dtype(y), B, C, H, W, _lib.stream(x)),
               "et_pack_input")
    return y


def maxpool5_fwd(x, out=None):
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((N, H, W, C), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().et_maxpool5_fwd(_lib.ptr(x), _nhwc(x), _lib.ptr(out), _nhwc(out), _lib.ptr(idx),
                                           et_dtype(x), N, H, W, C, _lib.stream(x)), "et_maxpool5_fwd")
    return out, idx


def maxpool5_bwd(dy, idx, base=None, out=None):
    N, H, W, C = dy.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    ldb = _nhwc(base) if base is not None else 0
    _lib.check(_lib.load().et_maxpool5_bwd(_lib.ptr(dy), _nhwc(dy), _lib.ptr(idx), _lib.ptr(base), ldb, _lib.ptr(out),
                                           _nhwc(out), et_dtype(dy), N, H, W, C, _lib.stream(dy)), "et_maxpool5_bwd")
    return out


def upsample2x_fwd(x, out=None):
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().et_upsample2x_fwd(_lib.ptr(x), _nhwc(x), _lib.ptr(out), _nhwc(out), et_dtype(x), N, H, W, C,
                                             _lib.stream(x)), "et_upsample2x_fwd")
    return out


def upsample2x_bwd(dy, out=None, accumulate=False):
    """accumulate: out += (out already holds the gradient of another consumer of the upsampled tensor)"""
    N, H2, W2, C = dy.shape
    H, W = H2 // 2, W2 // 2
    if out is None:
        assert not accumulate
        out = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    assert out.shape == (N, H, W, C) and out.dtype == dy.dtype
    _lib.check(_lib.load().et_upsample2x_bwd(_lib.ptr(dy), _nhwc(dy), _lib.ptr(out), _nhwc(out), et_dtype(dy), N, H, W, C,
                                             int(bool(accumulate)), _lib.stream(dy)), "et_upsample2x_bwd")
    return out


# ---- pseudo labels / losses ---------------------------------------------------------------------------
def pseudo_label_transform(dets, counts, M_s, width, height, clip01=False):
    """dets (B,max_det,8) fp32, counts (B) int32, M_s (B,13) fp64 -> targets9 (B*max_det,9) fp64, valid uint8.
    clip01: LabelMatch's extra clip of the normalised xywh (utils/labelmatch.py:333)."""
    B, max_det, _ = dets.shape
    dev = dets.device
    M_s = M_s.to(device=dev, dtype=torch.float64).contiguous()
    t9 = torch.empty((B * max_det, 9), dtype=torch.float64, device=dev)
    valid = torch.empty((B * max_det,), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().et_pseudo_label_transform(_lib.ptr(dets), _lib.ptr(counts), _lib.ptr(M_s), B, max_det,
                                                     int(width), int(height), int(bool(clip01)), _lib.ptr(t9),
                                                     _lib.ptr(valid), _lib.stream(dets)), "et_pseudo_label_transform")
    return t9, valid


def strong_view_u8(weak, minv, lut, cutouts, flags, border_value=114):
    """weak (B,3,H,W) uint8 -> strong view (B,3,H,W) uint8: warp + colour LUTs + cutouts + flips in one pass (csrc/augment.hip)"""
    B, C, H, W = weak.shape
    assert weak.dtype == torch.uint8 and C == 3 and weak.is_contiguous()
    assert minv.dtype == torch.float64 and minv.shape == (B, 6) and flags.dtype == torch.int32 and flags.shape == (B, 3)
    assert cutouts.dtype == torch.int32 and cutouts.shape == (B, 32, 7) and (lut is None or (lut.dtype == torch.uint8 and lut.shape == (B, 3, 256)))
    out = torch.empty_like(weak)
    _lib.check(_lib.load().et_strong_view_u8(_lib.ptr(weak), _lib.ptr(out), B, H, W, _lib.ptr(minv.contiguous()),
                                             _lib.ptr(lut.contiguous()) if lut is not None else None, _lib.ptr(cutouts.contiguous()),
                                             _lib.ptr(flags.contiguous()), int(border_value), _lib.stream(weak)), "et_strong_view_u8")
    return out


def score_log_append(dets, counts, conf_log, cls_log, log_count):
    """append (conf, cls) of every NMS detection to the device log (LabelMatch, utils/labelmatch.py:279-287)"""
    B, max_det, _ = dets.shape
    assert conf_log.dtype == torch.float32 and cls_log.dtype == torch.int32 and log_count.dtype == torch.int64
    assert conf_log.numel() == cls_log.numel()
    _lib.check(_lib.load().et_score_log_append(_lib.ptr(dets), _lib.ptr(counts), B, max_det, _lib.ptr(conf_log),
                                               _lib.ptr(cls_log), _lib.ptr(log_count), conf_log.numel(),
                                               _lib.stream(dets)), "et_score_log_append")


class DeviceThresholds:
    """The per-class (low, high) pseudo-label thresholds of ComputeStudentMatchLoss as ONE persistent fp64 device tensor
    (2, nc) per loss object.  The trainer rewrites the host lists when LabelMatch adapts them (ssod_trainer.py:322-323);
    ``refresh`` then copies the new values INTO the same device memory (stream-ordered, in place), so a captured step graph
    -- which bakes in the tensor's address -- reads the new thresholds at its next replay and the buffer it points at is
    never freed or replaced (the value-keyed cache this replaces handed a captured graph a tensor that a later cache
    eviction could free)."""

    def __init__(self):
        self._dev = {}          # device -> [tensor (2, nc) fp64, snapshot of the host lists last uploaded]

    def refresh(self, low, high, dev):
        key = (tuple(float(v) for v in low), tuple(float(v) for v in high))
        st = self._dev.get(dev)
        if st is None:
            st = self._dev[dev] = [torch.empty((2, len(key[0])), dtype=torch.float64, device=dev), None]
        if st[1] != key:
            if st[0].shape[1] != len(key[0]):
                raise ValueError("the number of classes of the threshold lists changed")
            if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                # a pageable host-to-device copy would invalidate the capture with an opaque error: say what went wrong instead
                raise RuntimeError("pseudo-label thresholds changed inside a step-graph capture: refresh_thresholds() must run "
                                   "before the capture starts (trainer/graph_step.py::StepGraph.run does)")
            st[0].copy_(torch.tensor([key[0], key[1]], dtype=torch.float64))
            st[1] = key
        return st[0]


def select_targets(targets9, valid, thr_low, thr_high, nc, with_obj, thresholds=None):
    """(N,9) fp64 pseudo labels (+ optional valid mask) -> (N,8) fp32 target table for et_yolo_loss.
    thresholds: the caller's DeviceThresholds (persistent device copy of the two lists); without one a temporary is made."""
    N = targets9.shape[0]
    dev = targets9.device
    t9 = targets9.to(torch.float64).contiguous()
    thr = (thresholds if thresholds is not None else DeviceThresholds()).refresh(thr_low, thr_high, dev)
    lo, hi = thr[0], thr[1]
    table = torch.empty((N, 8), dtype=torch.float32, device=dev)
    if N == 0:                            # no pseudo label at all: an empty table (the loss then has its objectness term only)
        return table
    _lib.check(_lib.load().et_select_targets(_lib.ptr(t9), _lib.ptr(valid), N, _lib.ptr(lo), _lib.ptr(hi), nc,
                                             int(bool(with_obj)), _lib.ptr(table), _lib.stream(t9)),
               "et_select_targets")
    return table


def _flat_span(p):
    """Elements of storage spanned by a strided 5-d logits view, rounded up to whole pixels."""
    span = sum((s - 1) * st for s, st in zip(p.shape, p.stride())) + 1
    sx = p.stride(3)
    return ((span + sx - 1) // sx) * sx if sx > 0 else span


def _loss_desc(p, table, anchors_host, balance, hp, pass_mask, ignore_obj):
    B, na = p[0].shape[0], p[0].shape[1]
    nc = hp["nc"]
    d = _lib.LossDesc()
    d.dtype = et_dtype(p[0]); d.B = B; d.na = na; d.nc = nc; d.NT = int(table.shape[0]); d.nl = len(p)
    for k in ("anchor_t", "gr", "cp", "cn", "cls_pw", "obj_pw", "box_w", "obj_w", "cls_w"):
        setattr(d, k, hp[k])
    d.pass_mask = pass_mask; d.ignore_obj = int(bool(ignore_obj))
    d.ota_match = None; d.obj_channel = 0; d.fl_gamma = float(hp.get("fl_gamma", 0.0))
    d.balance_dev = None; d.autobalance_ssi = -1
    for i, pi in enumerate(p):
        assert pi.dim() == 5 and pi.stride(4) == 1 and pi.shape[4] == nc + 5 and pi.dtype == p[0].dtype
        L = d.level[i]
        L.p = _lib.ptr(pi)
        L.sb, L.sa, L.sy, L.sx = pi.stride(0), pi.stride(1), pi.stride(2), pi.stride(3)
        L.ny, L.nx = pi.shape[2], pi.shape[3]
        for a in range(na):
            L.anchors[2 * a] = float(anchors_host[i][a][0]); L.anchors[2 * a + 1] = float(anchors_host[i][a][1])
        L.balance = float(balance[i])
    return d


def yolo_loss(p, table, anchors_host, balance, *, nc, anchor_t, gr, cp, cn, cls_pw, obj_pw, box_w, obj_w, cls_w,
              pass_mask=1, ignore_obj=False, ota_match=None, obj_channel=0, dps=None, fl_gamma=0.0, balance_dev=None, ssi=-1):
    """Fused assignment + loss + gradient.  p: list of (B,na,ny,nx,no) logits views (channel stride 1).
    Returns out (8,) fp32 [lbox, lobj, lcls, loss*bs, npos0..3] and the flat fp32 gradient buffers.
    ota_match / obj_channel: the SimOTA half of ComputeLoss.ota_loss (positives from `ota_assign`, objectness on another
    channel); dps: gradient buffers of an earlier call to accumulate into."""
    import ctypes
    lib = _lib.load()
    dev = p[0].device
    B, na = p[0].shape[0], p[0].shape[1]
    table = table.contiguous()
    hp = dict(nc=nc, anchor_t=anchor_t, gr=gr, cp=cp, cn=cn, cls_pw=cls_pw, obj_pw=obj_pw, box_w=box_w, obj_w=obj_w, cls_w=cls_w,
              fl_gamma=fl_gamma)
    d = _loss_desc(p, table, anchors_host, balance, hp, pass_mask, ignore_obj)
    acc = torch.empty(64, dtype=torch.float32, device=dev)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    d.targets = _lib.ptr(table) if table.numel() else _lib.ptr(acc)
    d.acc_ws = _lib.ptr(acc); d.out = _lib.ptr(out)
    if ota_match is not None:
        assert ota_match.dtype == torch.int32 and ota_match.numel() == len(p) * 5 * na * table.shape[0]
        d.ota_match = _lib.ptr(ota_match)
    d.obj_channel = int(obj_channel)
    if balance_dev is not None:          # Loss.autobalance: the weights live (and are updated) on the device
        assert balance_dev.dtype == torch.float32 and balance_dev.numel() >= len(p) and balance_dev.device == dev
        d.balance_dev = _lib.ptr(balance_dev); d.autobalance_ssi = int(ssi)
    keep, new_dps = [table, acc], []
    for i, pi in enumerate(p):
        _, _, ny, nx, _ = pi.shape
        dp = dps[i] if dps is not None else torch.zeros(_flat_span(pi), dtype=torch.float32, device=dev)
        assert dp.numel() == _flat_span(pi)
        tobj = torch.empty(B * na * ny * nx, dtype=torch.int64, device=dev)
        L = d.level[i]
        L.dp = _lib.ptr(dp); L.tobj_ws = _lib.ptr(tobj)
        new_dps.append(dp); keep.append(tobj)
    _lib.check(lib.et_yolo_loss(ctypes.byref(d), _lib.stream(p[0])), "et_yolo_loss")
    return out, new_dps


def ota_assign(p, table, anchors_host, strides, *, nc, anchor_t, top_k=13, img_size=640.0):
    """SimOTA dynamic-k matching on the device (build_ota_targets, yolo_anchor_assigner.py:104-264).
    Returns match (nl * 5*na*NT,) int32: matched row of `table` per candidate slot, -1 = not a positive."""
    import ctypes
    lib = _lib.load()
    dev = p[0].device
    B, na, nl, NT = p[0].shape[0], p[0].shape[1], len(p), int(table.shape[0])
    table = table.contiguous()
    hp = dict(nc=nc, anchor_t=anchor_t, gr=1.0, cp=1.0, cn=0.0, cls_pw=1.0, obj_pw=1.0, box_w=0.0, obj_w=0.0, cls_w=0.0)
    d = _loss_desc(p, table, anchors_host, [0.0] * nl, hp, 1, False)
    match = torch.full((nl * 5 * na * NT,), -1, dtype=torch.int32, device=dev)
    if NT == 0:
        return match
    d.targets = _lib.ptr(table)
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.et_ota_workspace_bytes(B, na, nl, NT, ctypes.byref(nbytes)), "et_ota_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    st = (ctypes.c_float * nl)(*[float(s) for s in strides])
    _lib.check(lib.et_ota_assign(ctypes.byref(d), st, float(img_size), int(top_k), _lib.ptr(ws), _lib.ptr(match),
                                 _lib.stream(p[0])), "et_ota_assign")
    return match


def scale_cast(src_flat, dtype, scale=1.0, dev_scale=None, out=None):
    dst = out if out is not None else torch.empty(src_flat.shape, dtype=dtype, device=src_flat.device)
    assert dst.dtype == dtype and dst.numel() == src_flat.numel() and dst.is_contiguous()
    _lib.check(_lib.load().et_scale_cast(_lib.ptr(src_flat), _lib.ptr(dst), et_dtype(dst), src_flat.numel(), float(scale),
                                         _lib.ptr(dev_scale), _lib.stream(src_flat)), "et_scale_cast")
    return dst


# ---- flat-arena state updates ------------------------------------------------------------------------
def cast_f32_to_lp(src, dst):
    """fp32 arena -> its 16-bit shadow in the compute format (dst.dtype bfloat16 | float16), round to nearest even"""
    assert src.dtype == torch.float32 and dst.dtype in LP_DTYPES and src.numel() == dst.numel()
    _lib.check(_lib.load().et_cast_f32_to_lp(_lib.ptr(src), _lib.ptr(dst), et_dtype(dst), src.numel(), _lib.stream(src)),
               "et_cast_f32_to_lp")
    return dst


def scaler_check(grads, scaler):
    """found_inf (scaler[2]) = 1 if any element of the scaled fp32 gradient arena is inf / nan"""
    assert grads.dtype == torch.float32 and scaler.dtype == torch.float32 and scaler.numel() >= 4
    _lib.check(_lib.load().et_scaler_check(_lib.ptr(grads), grads.numel(), _lib.ptr(scaler), _lib.stream(grads)), "et_scaler_check")


def scaler_update(scaler, growth_factor, backoff_factor, growth_interval):
    _lib.check(_lib.load().et_scaler_update(_lib.ptr(scaler), float(growth_factor), float(backoff_factor), int(growth_interval),
                                            _lib.stream(scaler)), "et_scaler_update")


def bn_eval_affine_into(gamma, beta, running_mean, running_var, eps, scale, shift):
    C = gamma.numel()
    _lib.check(_lib.load().et_bn_eval_affine(C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                             _lib.ptr(running_var), eps, _lib.ptr(scale), _lib.ptr(shift),
                                             _lib.stream(gamma)), "et_bn_eval_affine")


def ema_update(ema_flat, model_flat, d):
    """v = v*d ; v += (1-d)*m over a flat fp32 arena (utils/torch_utils.py:335-338)."""
    assert ema_flat.numel() == model_flat.numel() and ema_flat.dtype == torch.float32
    _lib.check(_lib.load().et_ema_update(_lib.ptr(ema_flat), _lib.ptr(model_flat), ema_flat.numel(), float(d),
                                         float(1. - d), _lib.stream(ema_flat)), "et_ema_update")


def ema_update_dev(ema_flat, model_flat, d2):
    """the same with d2 = device tensor [d, 1-d] (graph-replayable form)"""
    assert ema_flat.numel() == model_flat.numel() and ema_flat.dtype == torch.float32 and d2.dtype == torch.float32
    _lib.check(_lib.load().et_ema_update_dev(_lib.ptr(ema_flat), _lib.ptr(model_flat), ema_flat.numel(), _lib.ptr(d2),
                                             _lib.stream(ema_flat)), "et_ema_update_dev")


def _shadow_dt(shadow):
    return et_dtype(shadow) if shadow is not None else ET_BF16


def sgd_nesterov_dev(p, g, buf, shadow, hp, first_step, scaler=None):
    """hp: device tensor [lr, momentum, weight_decay, inv_scale] (graph-replayable form); scaler: the device loss-scaler state of
    fp16 mode (found_inf skips the update, 1/scale multiplies the gradient) or None"""
    assert hp.dtype == torch.float32 and hp.numel() >= 4
    _lib.check(_lib.load().et_sgd_nesterov_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(buf), _lib.ptr(shadow), _shadow_dt(shadow), p.numel(),
                                               _lib.ptr(hp), int(bool(first_step)), _lib.ptr(scaler), _lib.stream(p)), "et_sgd_nesterov_dev")


def adamw(p, g, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, weight_decay, step, inv_scale=1.0, scaler=None):
    _lib.check(_lib.load().et_adamw(_lib.ptr(p), _lib.ptr(g), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(shadow), _shadow_dt(shadow),
                                    p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                    float(inv_scale), _lib.ptr(scaler), _lib.stream(p)), "et_adamw")


def adamw_tick(tick, beta1, beta2, scaler):
    """advance AdamW's device-resident step count {t, 1 - b1^t, 1 - b2^t} (float64[3]) unless the scaler's found_inf is set"""
    assert tick.dtype == torch.float64 and tick.numel() == 3
    _lib.check(_lib.load().et_adamw_tick(_lib.ptr(tick), float(beta1), float(beta2), _lib.ptr(scaler), _lib.stream(tick)), "et_adamw_tick")


def adamw_dev(p, g, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, weight_decay, tick, inv_scale=1.0, scaler=None):
    _lib.check(_lib.load().et_adamw_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(shadow), _shadow_dt(shadow),
                                        p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), _lib.ptr(tick),
                                        float(inv_scale), _lib.ptr(scaler), _lib.stream(p)), "et_adamw_dev")


def sgd_nesterov(p, g, buf, shadow, lr, momentum, weight_decay, first_step, inv_scale=1.0, scaler=None):
    _lib.check(_lib.load().et_sgd_nesterov(_lib.ptr(p), _lib.ptr(g), _lib.ptr(buf), _lib.ptr(shadow), _shadow_dt(shadow), p.numel(),
                                           float(lr), float(momentum), float(weight_decay), int(bool(first_step)),
                                           float(inv_scale), _lib.ptr(scaler), _lib.stream(p)), "et_sgd_nesterov")


def detect_decode(raw5, anchor_px, stride, z, a_offset):
    """raw5: (B,na,ny,nx,no) logits view; writes z[:, a_offset : a_offset+na*ny*nx, :] (fp32)."""
    B, na, ny, nx, no = raw5.shape
    assert raw5.stride(4) == 1 and z.is_contiguous() and z.dtype == torch.float32
    _lib.check(_lib.load().et_detect_decode(_lib.ptr(raw5), et_dtype(raw5), B, na, ny, nx, no, raw5.stride(0),
                                            raw5.stride(1), raw5.stride(2), raw5.stride(3), _lib.ptr(anchor_px),
                                            float(stride), _lib.ptr(z), z.shape[1], a_offset, _lib.stream(z)),
               "et_detect_decode")


# ---- domain adaptation branch ----------------------------------------------------------------------------
def domain_focal(feat_nhwc, label, gscale, loss_sum, want_grad=True):
    """feat (B,H,W,CP) netD output (channels 0,1 = logits); accumulates the focal sum into loss_sum[0] and
    returns the gradient buffer (same shape, zeros outside channels 0,1) scaled by gscale."""
    N, H, W, CP = feat_nhwc.shape
    g = torch.zeros_like(feat_nhwc) if want_grad else None
    _lib.check(_lib.load().et_domain_focal(_lib.ptr(feat_nhwc), _nhwc(feat_nhwc), et_dtype(feat_nhwc), N * H * W, int(label),
                                           float(gscale), _lib.ptr(g), _nhwc(g) if g is not None else 0,
                                           _lib.ptr(loss_sum), _lib.stream(feat_nhwc)), "et_domain_focal")
    return g


def scale_inplace(x, alpha, dev_scale=None):
    assert x.is_contiguous()
    _lib.check(_lib.load().et_scale_inplace(_lib.ptr(x), et_dtype(x), x.numel(), float(alpha), _lib.ptr(dev_scale),
                                            _lib.stream(x)), "et_scale_inplace")
    return x


# ---- YOLOv8 anchor-free head ---------------------------------------------------------------------------------
def v8_decode(reg_nhwc, cls_nhwc, reg_max, nc, stride, cell_offset, out, a_offset):
    """one level of YoloV8Detect's inference decode: writes out[:, a_offset : a_offset + H*W, :] (B, A_total, 5+nc) fp32"""
    B, H, W, _ = reg_nhwc.shape
    assert cls_nhwc.shape[:3] == (B, H, W) and out.is_contiguous() and out.dtype == torch.float32 and reg_nhwc.dtype == cls_nhwc.dtype
    _lib.check(_lib.load().et_v8_decode(_lib.ptr(reg_nhwc), _nhwc(reg_nhwc), _lib.ptr(cls_nhwc), _nhwc(cls_nhwc), et_dtype(reg_nhwc),
                                        B, H, W, int(reg_max), int(nc), float(stride), float(cell_offset), _lib.ptr(out), out.shape[1],
                                        int(a_offset), _lib.stream(out)), "et_v8_decode")


def tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk=13, alpha=1.0, beta=6.0, eps=1e-9, return_idx=False):
    """TaskAlignedAssigner.forward on the device: -> (target_labels (B,A) int64, target_bboxes (B,A,4), target_scores (B,A,nc),
    fg_mask (B,A) bool [, gt_idx (B,A) int32: the owning gt where fg_mask is set])."""
    import ctypes
    B, A, nc = pd_scores.shape
    G = gt_bboxes.shape[1]
    dev = pd_scores.device
    f = lambda t: t.to(torch.float32).contiguous()
    tl = torch.empty((B, A), dtype=torch.int64, device=dev)
    tb = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    ts = torch.empty((B, A, nc), dtype=torch.float32, device=dev)
    fg = torch.empty((B, A), dtype=torch.uint8, device=dev)
    if G == 0:                                       # tal_assigner.py:52-57
        r = (tl.fill_(nc), tb.zero_(), ts.zero_(), fg.zero_().bool())
        return r + (torch.zeros((B, A), dtype=torch.int32, device=dev),) if return_idx else r
    lib = _lib.load()
    n = ctypes.c_size_t()
    _lib.check(lib.et_tal_assign_workspace_bytes(B, A, G, ctypes.byref(n)), "et_tal_assign_workspace_bytes")
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    ps, pb, ap = f(pd_scores), f(pd_bboxes), f(anc_points)
    gl, gb, gm = f(gt_labels.reshape(B, G)), f(gt_bboxes), f(mask_gt.reshape(B, G))
    _lib.check(lib.et_tal_assign(_lib.ptr(ps), _lib.ptr(pb), _lib.ptr(ap), _lib.ptr(gl), _lib.ptr(gb), _lib.ptr(gm), B, A, G, nc,
                                 int(topk), float(alpha), float(beta), float(eps), _lib.ptr(tl), _lib.ptr(tb), _lib.ptr(ts),
                                 _lib.ptr(fg), _lib.ptr(ws), n.value, _lib.stream(ps)), "et_tal_assign")
    if return_idx:
        idx = torch.empty((B, A), dtype=torch.int32, device=dev)
        _lib.check(lib.et_tal_assigned_gt(_lib.ptr(ws), B, A, G, _lib.ptr(idx), _lib.stream(ps)), "et_tal_assigned_gt")
        return tl, tb, ts, fg.bool(), idx
    return tl, tb, ts, fg.bool()


TAL_PAD_SYNC_FREE_ROWS = 32     # up to this many target rows per image ON AVERAGE the padded table is sized by n without a host sync


def mosaic4_u8(tiles, layouts, s, border_value=114):
    """B mosaics of four uint8 (3, h, w) images each (utils/augment.py MosaicGenerator): tiles[b] = the four device tensors in tile
    order, layouts[b] = their mosaic_layout rows.  -> (B, 3, s, s) uint8: the 2:1 box average of the (never materialised) 2s x 2s canvas"""
    B = len(tiles)
    dev = tiles[0][0].device
    table = torch.empty((B, 4, 8), dtype=torch.int64)
    keep = []
    for b in range(B):
        for i, (t, row) in enumerate(zip(tiles[b], layouts[b])):
            assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[0] == 3 and t.is_contiguous() and t.device == dev
            x1a, y1a, x2a, y2a, x1b, y1b = (int(v) for v in row[:6])
            table[b, i] = torch.tensor([t.data_ptr(), t.shape[1], t.shape[2], x1a, y1a, x2a, y2a, (x1b << 32) | (y1b & 0xffffffff)])
            keep.append(t)
    tdev = table.to(dev)
    out = torch.empty((B, 3, s, s), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().et_mosaic4_u8(_lib.ptr(tdev), _lib.ptr(out), B, int(s), int(border_value), _lib.stream(out)), "et_mosaic4_u8")
    return out


def tal_targets_pad(targets, B, img_w, img_h):
    """ComputeTalLoss.preprocess on the device: (n,6) [img, cls, x, y, w, h] normalised -> gt_labels (B,G,1), gt_bboxes (B,G,4) xyxy
    pixels, mask_gt (B,G,1).  G = max(n, 1) without a host synchronisation while n <= 32 * B (the reference loops over
    targets.cpu()); beyond that G = the largest per-image count (one host read)"""
    t = targets[:, :6].to(torch.float32).contiguous()
    n = int(t.shape[0])
    G = max(n, 1)
    if n > TAL_PAD_SYNC_FREE_ROWS * B:
        # crowded batch (mosaic: thousands of boxes): G = n would make the assigner's B*G*A workspace and its B*G workgroups grow with the
        # BATCH total instead of the per-image maximum the reference pads to (tal_loss.py:131-143).  One host read of the per-image
        # maximum -- the reference's own preprocess synchronises on targets.cpu() at this point anyway.
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            # a host read is illegal inside a graph capture (it would invalidate the capture and the trainer would silently fall back
            # to eager steps for the rest of the run -- ADVICE r04), and a data-dependent G could not be replayed anyway
            raise RuntimeError(f"tal_targets_pad: {n} targets for {B} images (> {TAL_PAD_SYNC_FREE_ROWS} per image on average) needs a "
                               "host read of the per-image maximum, which a step-graph capture cannot contain: run this batch "
                               "eagerly (ET_STEP_GRAPH=0) or cap the targets per image in the loader")
        G = max(int(torch.bincount(t[:, 0].long().clamp_(0, B - 1), minlength=B).max()), 1)
    out = torch.empty((B, G, 5), dtype=torch.float32, device=t.device)
    mask = torch.empty((B, G, 1), dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().et_tal_targets_pad(_lib.ptr(t) if n else None, n, B, G, float(img_w), float(img_h), _lib.ptr(out), _lib.ptr(mask),
                                              _lib.stream(out)), "et_tal_targets_pad")
    return out[..., :1], out[..., 1:], mask


def tal_pseudo_split(targets9, valid, thr, B, nc, with_obj, with_bbox, with_cls, img_w, img_h):
    """padded pseudo labels (B*G, 9) fp64 (+ valid) and thr = (2, nc) fp64 [low; high] -> per set (reliable, uncertain) a padded gt
    table (labels (B,G,1), boxes (B,G,4) xyxy px, mask (B,G,1)), then u_score (B,G), u_flags (B,G) uint8  (EXTENSION: include/et_hip.h)"""
    n = targets9.shape[0]
    assert n % B == 0 and targets9.dtype == torch.float64 and targets9.is_contiguous()
    G = n // B
    dev = targets9.device
    f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    glr, gbr, glu, gbu, mr, mu, us = f32(B, G, 1), f32(B, G, 4), f32(B, G, 1), f32(B, G, 4), f32(B, G, 1), f32(B, G, 1), f32(B, G)
    uf = torch.empty((B, G), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().et_tal_pseudo_split(_lib.ptr(targets9), _lib.ptr(valid), _lib.ptr(thr[0]), _lib.ptr(thr[1]), B, G, nc,
                                               int(bool(with_obj)), int(bool(with_bbox)), int(bool(with_cls)), float(img_w), float(img_h),
                                               _lib.ptr(glr), _lib.ptr(gbr), _lib.ptr(glu), _lib.ptr(gbu), _lib.ptr(mr), _lib.ptr(mu),
                                               _lib.ptr(us), _lib.ptr(uf), _lib.stream(targets9)), "et_tal_pseudo_split")
    return (glr, gbr, mr), (glu, gbu, mu), us, uf


def tal_merge_pseudo(rel, unc, u_score, u_flags):
    """rel = (tb, ts, fg) of the reliable labels, unc = (tb, ts, fg, gt_idx) of the uncertain ones -> merged (target_bboxes,
    target_scores, fg_box uint8): the uncertain owner of an anchor wins (ssod_loss.py:231 then :248)"""
    tb_r, ts_r, fg_r = rel
    tb_u, ts_u, fg_u, idx_u = unc
    B, A, nc = ts_r.shape
    G = u_score.shape[1]
    dev = ts_r.device
    ts, tb = torch.empty_like(ts_r), torch.empty_like(tb_r)
    fg = torch.empty((B, A), dtype=torch.uint8, device=dev)
    fr8, fu8 = fg_r.to(torch.uint8).contiguous(), fg_u.to(torch.uint8).contiguous()     # named: they must outlive the launch
    _lib.check(_lib.load().et_tal_merge_pseudo(_lib.ptr(ts_r), _lib.ptr(tb_r), _lib.ptr(fr8), _lib.ptr(ts_u), _lib.ptr(tb_u),
                                               _lib.ptr(fu8), _lib.ptr(idx_u), _lib.ptr(u_score), _lib.ptr(u_flags), B, A, G, nc,
                                               _lib.ptr(ts), _lib.ptr(tb), _lib.ptr(fg), _lib.stream(ts_r)), "et_tal_merge_pseudo")
    return tb, ts, fg


def tal_loss(pred_scores, pred_distri, anchor_points_s, stride_tensor, target_bboxes_px, target_scores, fg_mask, reg_max, iou_type,
             w_class, w_iou, w_dfl):
    """fused ComputeTalLoss terms + gradients (et_tal_loss): -> out (4,) [iou, dfl, cls, total] weighted, grad_scores, grad_distri"""
    B, A, nc = pred_scores.shape
    dev = pred_scores.device
    kind = {"iou": 0, "giou": 1}.get(iou_type)
    if kind is None:
        raise NotImplementedError(f"Loss.iou_type {iou_type}: the fused TAL loss implements 'giou' (default) and 'iou'")
    f = lambda t: t.to(torch.float32).contiguous()
    ps, pd = f(pred_scores), f(pred_distri)
    gs, gd = torch.empty_like(ps), torch.empty_like(pd)
    acc = torch.zeros(4, dtype=torch.float32, device=dev)
    out = torch.empty(4, dtype=torch.float32, device=dev)
    fg8 = fg_mask.to(torch.uint8).contiguous()
    aps, stv, tbp, tsc = f(anchor_points_s), f(stride_tensor.reshape(-1)), f(target_bboxes_px), f(target_scores)   # named: alive across the launch
    _lib.check(_lib.load().et_tal_loss(_lib.ptr(ps), _lib.ptr(pd), _lib.ptr(aps), _lib.ptr(stv),
                                       _lib.ptr(tbp), _lib.ptr(tsc), _lib.ptr(fg8), B, A, nc, int(reg_max),
                                       kind, float(w_class), float(w_iou), float(w_dfl), _lib.ptr(gs), _lib.ptr(gd), _lib.ptr(acc),
                                       _lib.ptr(out), _lib.stream(ps)), "et_tal_loss")
    return out, gs, gd
