"""Data parallelism for the flat-arena model: one process per GPU, gradients summed with RCCL
all-reduce over the flat gradient arena (torch.distributed backend "nccl" == RCCL on ROCm; "gloo" on
CPU for the world_size-2 tests).

Replaces DistributedDataParallel as used at reference trainer/trainer.py:313 (bucket_cap 25 MB,
find_unused_parameters=True, broadcast_buffers=True).  Semantics kept: parameters (and buffers) are
broadcast from rank 0 at construction; after backward every rank holds the MEAN over ranks of the
local gradients -- the reference then multiplies the loss by WORLD_SIZE (trainer.py:425-426), i.e. the
effective gradient is the SUM over ranks of per-rank mean-losses, and that convention is preserved.
Because all gradients already live in one contiguous fp32 arena there is nothing to bucket or to
traverse: the arena is cut into a few large chunks sized for xGMI's per-link bandwidth and each chunk
is all-reduced asynchronously as soon as backward has passed the layers it covers.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


def apply_rccl_knobs(env=None):
    """Tuning knobs of the collective library, to be called BEFORE ``init_process_group`` (RCCL reads its environment when
    the communicator is created).  ``ET_RCCL_CHANNELS=n`` pins the number of RCCL channels (= workgroups = CUs the
    all-reduce kernel occupies): the gradient all-reduce runs beside backward's 256x256-tile kernels, which want every CU,
    so the CU share of the collective is a trade between its own bandwidth (7 xGMI links x ~153 GB/s per GPU, a ring is
    per-link bound) and the compute it displaces.  (Protocol / algorithm: set RCCL's own NCCL_PROTO / NCCL_ALGO.)
    Values already present in the environment win.  Returns what was set (bench.py records it)."""
    env = os.environ if env is None else env
    out = {}
    n = env.get("ET_RCCL_CHANNELS")
    if n:
        for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS"):
            if k not in env:
                env[k] = str(int(n))
            out[k] = env[k]
    for k in ("NCCL_PROTO", "NCCL_ALGO"):           # recorded when the user set them
        if env.get(k):
            out[k] = env[k]
    return out


class FlatDataParallel(nn.Module):
    def __init__(self, module, process_group=None, chunk_mb=None, broadcast_buffers=True, overlap=True, single_rank_collectives=None,
                 grad_dtype=None):
        """chunk_mb: size of the all-reduce pieces of the conv-weight gradient segment (default 48, ``ET_ALLREDUCE_CHUNK_MB``).
        grad_dtype: None / torch.float32 = the fp32 gradient arena is reduced as it is (191.8 MB per step for YOLOv5l);
        torch.bfloat16 = every piece is cast into a persistent bf16 staging arena, reduced there (half the bytes on the xGMI
        links: a ring all-reduce is per-link bound) and cast back into the fp32 arena once its collective has completed.  The
        mean then carries bf16 rounding (2^-9 relative per element, bounded in tests/test_parallel.py); master weights, momentum
        and the optimizer stay fp32.  Opt-in: the reference reduces fp32 gradients (trainer/trainer.py:313, DDP).
        single_rank_collectives (``ET_DP_SINGLE_RANK=1``): issue every collective even in a group of ONE rank -- the way to
        execute the RCCL code path (AVG all-reduce from the gradient-ready hook, broadcast, capture into a step graph) on a
        single-GPU box; with more ranks it changes nothing."""
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if chunk_mb is None:
            chunk_mb = float(os.environ.get("ET_ALLREDUCE_CHUNK_MB", "48"))
        self.chunk = int(chunk_mb * (1 << 20) // 4)
        if grad_dtype is None:                       # transport setting, like the chunk size: ET_ALLREDUCE_DTYPE=bf16 | fp32 (default)
            env = os.environ.get("ET_ALLREDUCE_DTYPE", "").strip().lower()
            if env not in ("", "fp32", "bf16"):
                raise ValueError(f"ET_ALLREDUCE_DTYPE={env!r}: expected 'fp32' (default) or 'bf16'")
            grad_dtype = torch.bfloat16 if env == "bf16" else torch.float32
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"grad_dtype {grad_dtype}: float32 (default) or bfloat16")
        self.grad_dtype = grad_dtype
        self._stage = None               # bf16 staging arena, same element offsets as the gradient arena (grad_dtype = bfloat16)
        self.broadcast_buffers = broadcast_buffers
        self.overlap = overlap
        if single_rank_collectives is None:
            single_rank_collectives = os.environ.get("ET_DP_SINGLE_RANK", "0") == "1"
        # `active`: collectives are issued (more than one rank, or the single-rank test mode with an initialised group)
        self.active = self.world > 1 or (bool(single_rank_collectives) and dist.is_initialized())
        self._works = []
        self._launched = set()
        self._dirty = False              # a backward has moved the per-chunk counters since the last reduce_gradients()
        self._dirty_gen = -1             # FlatState.zero_gen when that backward started
        self._stale_gen = None           # set by forward() after an aborted backward whose partial gradients are still in the arena
        self.timing = False              # bench.py: HIP events around the collective phase of a step
        self.last_timing = None          # (first launch -> all complete, exposed wait after backward) in ms
        self._ev0 = None
        self._pending_timing = None
        if self.active:
            f = module.flat_state()
            dist.broadcast(f.params, 0, group=self.pg)
            dist.broadcast(f.buffers, 0, group=self.pg)
            f.mark_weights_changed()
            self._plan()
            if overlap:
                from . import autograd as _ag
                _ag.GRAD_READY_HOOK = self._on_conv_grad_ready

    # ---- chunk plan: the conv-weight segment is cut into ~chunk_mb pieces on layer boundaries --------------
    def _plan(self):
        f = self.module.flat_state()
        wo, wn = f.w_range
        slots = sorted(f.conv_slots.values(), key=lambda s: s.index)
        base = f.grads.data_ptr()
        self._chunks = []            # [offset, numel, remaining_layers]
        self._slot_chunk = {}
        cur_o, cur_n, members = None, 0, []
        for s in slots:
            o = (s.gw.data_ptr() - base) // 4
            if cur_o is None:
                cur_o = o
            members.append(s.index)
            cur_n = o + s.gw.numel() - cur_o
            if cur_n >= self.chunk:
                self._chunks.append([cur_o, cur_n, len(members)])
                for m in members:
                    self._slot_chunk[m] = len(self._chunks) - 1
                cur_o, cur_n, members = None, 0, []
        if members:
            self._chunks.append([cur_o, wo + wn - cur_o, len(members)])
            for m in members:
                self._slot_chunk[m] = len(self._chunks) - 1
        if self._chunks:                      # cover alignment padding between layers / at the segment end
            self._chunks[0][1] += self._chunks[0][0] - wo
            self._chunks[0][0] = wo
            for i in range(len(self._chunks) - 1):
                self._chunks[i][1] = self._chunks[i + 1][0] - self._chunks[i][0]
            self._chunks[-1][1] = wo + wn - self._chunks[-1][0]
        self._remaining = [c[2] for c in self._chunks]

    def _all_reduce(self, view):
        if self.timing and self._ev0 is None and view.is_cuda:
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev0.record()
        avg = dist.get_backend(self.pg) == "nccl"        # RCCL averages inside the collective
        wire, back = view, None
        if self.grad_dtype != torch.float32:
            g = self.module.flat_state().grads
            if self._stage is None or self._stage.numel() != g.numel() or self._stage.device != g.device:
                self._stage = torch.empty(g.numel(), dtype=self.grad_dtype, device=g.device)
            o = (view.data_ptr() - g.data_ptr()) // 4
            wire = self._stage[o:o + view.numel()]
            wire.copy_(view)                             # on the launching stream, in front of the collective (async_op orders it behind)
            back = view
        w = dist.all_reduce(wire, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._works.append((w, None if avg else wire, int(wire.numel()) * wire.element_size(), wire, back))

    def _on_conv_grad_ready(self, slot):
        """Called from the conv backward right after its wgrad launch: when every layer of a chunk has
        produced its gradient, the chunk's all-reduce starts while backward continues upstream."""
        ci = self._slot_chunk.get(slot.index)
        if ci is None or ci in self._launched:
            return
        if not self._dirty:
            self._dirty_gen = self.module.flat_state().zero_gen
        self._dirty = True                    # counters no longer at their start values (cleared by reduce_gradients)
        self._remaining[ci] -= 1
        if self._remaining[ci] == 0:
            o, n, _ = self._chunks[ci]
            self._launched.add(ci)
            self._all_reduce(self.module.flat_state().grads[o:o + n])

    def forward(self, *a, **k):
        if self._works or self._launched:
            # a backward whose collectives nobody finished (an exception between backward and reduce_gradients, or a caller
            # that skipped it): never let an all-reduce in flight overlap the next backward's writes into the same arena.
            # A chunk was launched, i.e. every layer of it had produced its gradient: the matching collectives exist on every rank
            # that ran the same backward, so finishing the set here pairs up.
            self.reduce_gradients()
        elif self._dirty:
            # a backward that aborted after some wgrad hooks had counted down but BEFORE any chunk was complete: nothing is in
            # flight, and issuing collectives from here would not pair up if the abort was local to this rank (the peers issue
            # none: a hang, or mismatched buffers).  Only the stale counters are reset -- with them the next backward would launch
            # a chunk's all-reduce before all of its layers had accumulated.  The half-accumulated gradient arena is the caller's
            # to zero (optimizer.zero_grad / FlatState.zero_grad), as after any failed step.
            import warnings
            warnings.warn("FlatDataParallel: the previous backward did not finish (no collective had been launched); "
                          "per-chunk counters reset, gradients of that pass are NOT reduced")
            # ... and unless the arena has been zeroed since that backward started, it still holds this rank's partial, UNREDUCED
            # gradients: the next reduce_gradients() refuses to average them into the step (ranks would diverge silently) until
            # zero_grad() has run
            stale = self.module.flat_state().zero_gen == self._dirty_gen
            self.abort_step()
            if stale:
                self._stale_gen = self._dirty_gen
        if self.active and self.broadcast_buffers and self.module.training:
            # DDP broadcast_buffers=True: rank 0's BN running stats / anchors at every forward, on the compute stream, in front of the
            # forward.  r04 moved it to a side stream (joined before the first running-statistics update) and took that back: the
            # forward's first launches already write the arena (the bulk num_batches_tracked bump of flat_state.prepare_forward), so the
            # broadcast raced with them on the non-root ranks; and with the collective on a forked stream the single-rank RCCL test
            # aborted in torch's NCCL watchdog (hipErrorCapturedEvent) once in three processes.  The broadcast is ~0.4 MB: tens of
            # microseconds per step on xGMI.
            dist.broadcast(self.module.flat_state().buffers, 0, group=self.pg)
        return self.module(*a, **k)

    def abort_step(self):
        """forget a step that will not be finished (a rejected graph capture): collectives already started are waited for,
        counters and flags return to their start-of-step values; the gradient arena is the caller's to zero"""
        for w, *_ in self._works:
            w.wait()
        self._works = []
        self._launched = set()
        self._remaining = [c[2] for c in self._chunks] if self.active else []
        self._dirty = False
        self._ev0 = None

    def reduce_gradients(self):
        """Finish the gradient all-reduce (mean over ranks); call after EVERY backward() -- also on the micro-steps of a
        gradient accumulation: the arena then holds (sum of the earlier, already averaged micro-gradients) + (this rank's
        local micro-gradient), and the mean over ranks of that is the sum of the averaged micro-gradients, exactly what
        DistributedDataParallel accumulates in .grad."""
        if not self.active:
            return
        if self._stale_gen is not None:
            if self.module.flat_state().zero_gen == self._stale_gen:
                raise RuntimeError("FlatDataParallel: the gradient arena still holds the unreduced partial gradients of an aborted "
                                   "backward (see the earlier warning): call zero_grad() and redo the step before reducing")
            self._stale_gen = None
        g = self.module.flat_state().grads
        wo, wn = self.module.flat_state().w_range
        for ci, (o, n, _) in enumerate(self._chunks):      # chunks whose layers did not all run (frozen / unused)
            if ci not in self._launched:
                self._all_reduce(g[o:o + n])
        self._all_reduce(g[:wo])                           # biases (BN + conv)
        self._all_reduce(g[wo + wn:])                      # BN weights
        timed = self.timing and self._ev0 is not None
        evs = []
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()                                     # backward is over on this stream: what follows is exposed
            evs.append(e)
        sizes = []
        for w, view, nbytes, wire, back in self._works:
            w.wait()
            if view is not None:                           # gloo (CPU tests) has no AVG
                view.mul_(1.0 / self.world)
            if back is not None:                           # bf16 wire format: the mean returns to the fp32 arena
                back.copy_(wire)
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()                                 # the compute stream has passed this collective's completion
                evs.append(e)
                sizes.append(nbytes)
        if timed:
            self._pending_timing = (self._ev0, evs, sizes)
            self._ev0 = None
        self._works = []
        self._launched = set()
        self._remaining = [c[2] for c in self._chunks]
        self._dirty = False

    def collect_timing(self):
        """(ms from the first all-reduce launch to the completion of the last, ms the compute stream waited for the
        collectives after backward, [(bytes, exposed ms) per collective in wait order]) of the last timed step;
        synchronises the events."""
        t = self._pending_timing
        if t is None:
            return None
        e0, evs, sizes = t
        evs[-1].synchronize()
        self._pending_timing = None
        per = [(sizes[i], evs[i].elapsed_time(evs[i + 1])) for i in range(len(sizes))]
        return e0.elapsed_time(evs[-1]), evs[0].elapsed_time(evs[-1]), per

    def flat_state(self):
        return self.module.flat_state()
