"""``DevicePrefetcher``: batches from pinned host memory into HBM on a copy stream, one batch ahead of the step.

The reference's DataLoader hands ``training_step`` host tensors and Lightning moves them to the GPU before the call
(``transfer_batch_to_gpu``), on the compute stream: the step waits for the whole copy.  The boundary of this build takes device
tensors; this small class is the piece in between for a caller that owns the loop: it keeps TWO device copies of a batch, fills the
one the step is not using on its own HIP stream (``non_blocking`` copies out of pinned memory: PCIe traffic beside compute), and
hands the step the other.  Events order the two streams in both directions -- the step waits for its batch's copy, the copy into a
slot waits until the step that read that slot has been enqueued completely -- and nothing synchronises the host.

Any nesting of tuples / lists / dicts of tensors is copied leaf by leaf (the collate's ``(sample tuple, target tuple, road_image
tuple)`` of helper.py:22-23 included); non-tensor leaves pass through.  uint8 camera frames ([6,H,W,3] per sample, what a JPEG
decoder emits) are 4x fewer PCIe bytes than ToTensor'd fp32 views and the models read them as they are (``ops.wide_image``).

    pf = DevicePrefetcher(loader, device)          # loader yields host batches (ideally pinned)
    for i, batch in enumerate(pf):                 # batch lives in HBM; valid until the next iteration has been requested
        step(batch, i)
"""
import torch


def _map(batch, fn):
    if isinstance(batch, torch.Tensor):
        return fn(batch)
    if isinstance(batch, tuple):
        return tuple(_map(b, fn) for b in batch)
    if isinstance(batch, list):
        return [_map(b, fn) for b in batch]
    if isinstance(batch, dict):
        return {k: _map(v, fn) for k, v in batch.items()}
    return batch


def _leaves(batch, out):
    if isinstance(batch, torch.Tensor):
        out.append(batch)
    elif isinstance(batch, (tuple, list)):
        for b in batch:
            _leaves(b, out)
    elif isinstance(batch, dict):
        for v in batch.values():
            _leaves(v, out)
    return out


class DevicePrefetcher:
    def __init__(self, batches, device, pin=False):
        """``batches``: an iterable of host batches.  ``pin=True`` pins each batch's tensors first when the producer did not (an
        extra host copy; a DataLoader with ``pin_memory=True`` makes it unnecessary)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher: the hot path runs on MI355X only")
        self.source = batches
        self.pin = pin
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [None, None]                              # device-side batch structures, reused when shapes repeat
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]  # copy into slot finished (recorded on the copy stream)
        self.freed = [torch.cuda.Event(), torch.cuda.Event()]  # the step that read the slot has been enqueued (compute stream)

    def _fetch(self, host, slot):
        if self.pin:
            host = _map(host, lambda t: t if t.is_pinned() or t.is_cuda else t.pin_memory())
        old = self.slots[slot]
        old_leaves = _leaves(old, []) if old is not None else []
        fresh = iter(old_leaves)

        def place(t):
            dst = next(fresh, None)
            if dst is None or dst.shape != t.shape or dst.dtype != t.dtype:
                dst = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            dst.copy_(t, non_blocking=True)
            return dst
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.freed[slot])
            self.slots[slot] = _map(host, place)
            self.ready[slot].record(self.copy_stream)

    def __iter__(self):
        it = iter(self.source)
        compute = torch.cuda.current_stream(self.device)
        for e in self.freed:
            e.record(compute)
        try:
            nxt = next(it)
        except StopIteration:
            return
        slot = 0
        self._fetch(nxt, slot)
        while True:
            try:
                nxt = next(it)
                more = True
            except StopIteration:
                more = False
            if more:
                self._fetch(nxt, slot ^ 1)                     # batch i+1 crosses PCIe while step i computes
            compute = torch.cuda.current_stream(self.device)
            compute.wait_event(self.ready[slot])
            yield self.slots[slot]
            self.freed[slot].record(torch.cuda.current_stream(self.device))      # everything the step enqueued reads the slot before this
            if not more:
                return
            slot ^= 1
