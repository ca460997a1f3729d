"""Host -> device staging of the loaders' batches one step ahead.

The reference moves every batch with ``imgs.to(device, non_blocking=True).float() / 255.0`` inside the step
(trainer/ssod_trainer.py:694-696): three uint8 -> fp32 conversions and three divisions over 3 x 32 x 3 x 640 x 640 elements
on the compute stream, after a copy the step has to wait for.  Here the uint8 batches stay uint8: they are copied on a
dedicated HIP copy stream while the previous step computes (pinned staging, 39 MB per 32-image batch), and the
normalisation happens inside the input pack kernel (et_pack_input_u8) -- the compute stream only waits on an event.
"""
import torch


class DevicePrefetcher:
    """Wraps an iterable of batches (tuples / lists whose tensor items are moved; other items pass through)."""

    def __init__(self, iterable, device, depth=2):
        self.it = iter(iterable)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.depth = max(1, depth)
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.queue = []
        self._pinned = {}

    def _stage(self, batch):
        if not self.cuda:
            return batch, None
        out = []
        with torch.cuda.stream(self.stream):
            for k, t in enumerate(batch):
                if torch.is_tensor(t) and not t.is_cuda:
                    key = (k, tuple(t.shape), t.dtype, len(self.queue) % (self.depth + 1))
                    buf = self._pinned.get(key)
                    if buf is None:
                        buf = self._pinned[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                    buf.copy_(t)
                    out.append(buf.to(self.device, non_blocking=True))
                else:
                    out.append(t)
            ev = torch.cuda.Event()
            ev.record(self.stream)
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
