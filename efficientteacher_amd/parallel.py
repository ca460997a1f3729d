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
import torch
import torch.distributed as dist
import torch.nn as nn


class FlatDataParallel(nn.Module):
    def __init__(self, module, process_group=None, chunk_mb=64, broadcast_buffers=True):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.chunk = int(chunk_mb * (1 << 20) // 4)
        self.broadcast_buffers = broadcast_buffers
        if self.world > 1:
            f = module.flat_state()
            dist.broadcast(f.params, 0, group=self.pg)
            dist.broadcast(f.buffers, 0, group=self.pg)
            f.mark_weights_changed()

    def forward(self, *a, **k):
        if self.world > 1 and self.broadcast_buffers and self.module.training:
            # DDP broadcast_buffers=True: rank 0's BN running stats / anchors at every forward
            dist.broadcast(self.module.flat_state().buffers, 0, group=self.pg)
        return self.module(*a, **k)

    def reduce_gradients(self):
        """All-reduce (mean) the flat gradient arena in large chunks; call after backward()."""
        if self.world <= 1:
            return
        g = self.module.flat_state().grads
        avg = dist.get_backend(self.pg) == "nccl"        # RCCL averages in the collective itself
        works = []
        for o in range(0, g.numel(), self.chunk):
            works.append(dist.all_reduce(g[o:o + self.chunk], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM,
                                         group=self.pg, async_op=True))
        for w in works:
            w.wait()
        if not avg:                                       # gloo (CPU tests) has no AVG
            g.mul_(1.0 / self.world)

    def flat_state(self):
        return self.module.flat_state()
