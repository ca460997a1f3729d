"""Host -> device staging of the loaders' batches one step ahead.

The reference moves every batch with ``imgs.to(device, non_blocking=True).float() / 255.0`` inside the step
(trainer/ssod_trainer.py:694-696): three uint8 -> fp32 conversions and three divisions over 3 x 32 x 3 x 640 x 640 elements
on the compute stream, after a copy the step has to wait for.  Here the uint8 batches stay uint8: they are copied on a
dedicated HIP copy stream while the previous step computes (pinned staging, 39 MB per 32-image batch), and the
normalisation happens inside the input pack kernel (et_pack_input_u8) -- the compute stream only waits on an event.
"""
import torch


class DevicePrefetcher:
    """Wraps an iterable of batches (tuples / lists whose tensor items are moved; other items pass through).

    Pinned staging: one ring of ``depth + 1`` SLOTS; batch number s (a monotonically increasing stage counter) uses slot
    ``s % (depth + 1)``.  A slot's pinned buffers are rewritten only after the event recorded behind the slot's previous
    host->device copies has completed (the copies are asynchronous: without that wait the host would overwrite pixels the
    DMA engine is still reading).  A buffer is a flat byte arena per (item index, slot) that grows to the largest item seen,
    so variable-length label tensors (n, 6) reuse one allocation instead of pinning a new one per distinct n."""

    def __init__(self, iterable, device, depth=2):
        self.it = iter(iterable)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.depth = max(1, depth)
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.queue = []
        self._slots = self.depth + 1
        self._stage_no = 0                 # batches staged so far
        self._pinned = {}                  # (item index, slot) -> flat uint8 pinned arena
        self._slot_ev = [None] * self._slots

    def _pinned_view(self, k, slot, t):
        nbytes = t.numel() * t.element_size()
        buf = self._pinned.get((k, slot))
        if buf is None or buf.numel() < nbytes:
            buf = self._pinned[(k, slot)] = torch.empty(max(nbytes, 16), dtype=torch.uint8).pin_memory()
        return buf[:nbytes].view(t.dtype).view(t.shape)

    def _stage(self, batch):
        if not self.cuda:
            return batch, None
        slot = self._stage_no % self._slots
        self._stage_no += 1
        if self._slot_ev[slot] is not None:        # the copies that last read this slot's pinned buffers are done
            self._slot_ev[slot].synchronize()
        out = []
        with torch.cuda.stream(self.stream):
            for k, t in enumerate(batch):
                if torch.is_tensor(t) and not t.is_cuda:
                    if t.numel() == 0:
                        out.append(torch.empty(t.shape, dtype=t.dtype, device=self.device))
                        continue
                    buf = self._pinned_view(k, slot, t)
                    buf.copy_(t)
                    out.append(buf.to(self.device, non_blocking=True))
                else:
                    out.append(t)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_ev[slot] = ev
        return out, ev

    def _fill(self):
        while len(self.queue) < self.depth:
            try:
                b = next(self.it)
            except StopIteration:
                return
            self.queue.append(self._stage(b))

    def __iter__(self):
        return self

    def __next__(self):
        self._fill()
        if not self.queue:
            raise StopIteration
        batch, ev = self.queue.pop(0)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in batch:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream(self.device))
        self._fill()                        # the NEXT batch starts moving while this one is being computed on
        return batch
