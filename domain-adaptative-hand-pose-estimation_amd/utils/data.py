"""Endless batch source over a finite loader (API of the reference's ``utils/data.py``:
``ForeverDataIterator(loader)`` with ``next()`` and ``len()``)."""
import itertools

import torch


def _cycle_fresh(loader):
    # itertools.cycle would replay the cached first epoch; a shuffling loader must be re-iterated instead
    passes = 0
    while True:
        empty = True
        sampler = getattr(loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):        # DistributedSampler: a new permutation for every pass over the shard
            sampler.set_epoch(passes)
        passes += 1
        for batch in loader:
            empty = False
            yield batch
        if empty:
            raise ValueError('data loader yields no batches')


class ForeverDataIterator:
    """Restarts the wrapped loader whenever it is exhausted, so training loops can draw a fixed number of
    iterations per epoch regardless of the dataset size."""

    def __init__(self, data_loader):
        self.data_loader = data_loader
        self._stream = _cycle_fresh(data_loader)

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._stream)

    def __len__(self):
        return len(self.data_loader)

    def take(self, n):
        """The next n batches as a list (convenience for warm-up / tests)."""
        return list(itertools.islice(self._stream, n))


class DevicePrefetcher:
    """Host -> HBM staging off the critical path (replaces the blocking ``.to(device)`` calls of the
    reference's loop, train1.py:359-366): while the training step consumes batch i, batch i+1 is copied on a side stream
    from pinned host memory.  Tensors the loader did not pin go through persistent pinned staging buffers (two slots,
    double-buffered), so no pinned allocation happens per batch.  Non-tensor items (the ``meta`` dicts) pass through.

    ``next()`` returns the batch with every tensor on ``device``; the consumer's stream waits on the copy event and the
    tensors are recorded on it, so the caching allocator does not recycle them early."""

    def __init__(self, iterator, device, slots=2):
        self.it = iter(iterator)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('DevicePrefetcher stages to a GPU; there is no CPU path')
        self.stream = torch.cuda.Stream(self.device)
        self.slots, self._slot = int(slots), 0
        self._staging = [dict() for _ in range(self.slots)]      # slot -> {(index, shape, dtype): pinned buffer}
        self._slot_free = [None] * self.slots                   # event: the slot's last copies have left host memory
        self._ready = None
        self._preload()

    def _to_device(self, slot, idx, t):
        if not t.is_pinned():
            key = (idx, tuple(t.shape), t.dtype)
            buf = self._staging[slot].get(key)
            if buf is None:
                buf = self._staging[slot][key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            buf.copy_(t)
            t = buf
        return t.to(self.device, non_blocking=True)

    def _preload(self):
        try:
            host = next(self.it)
        except StopIteration:
            self._ready = None
            return
        slot = self._slot
        self._slot = (slot + 1) % self.slots
        if self._slot_free[slot] is not None:
            self._slot_free[slot].synchronize()                 # the staging buffers of this slot are about to be rewritten
        seq = isinstance(host, (tuple, list))
        items = list(host) if seq else [host]
        with torch.cuda.stream(self.stream):
            out = [self._to_device(slot, i, x) if torch.is_tensor(x) else x for i, x in enumerate(items)]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_free[slot] = ev
        self._ready = (type(host)(out) if seq else out[0], ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._ready is None:
            raise StopIteration
        batch, ev = self._ready
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for x in (batch if isinstance(batch, (tuple, list)) else [batch]):
            if torch.is_tensor(x):
                x.record_stream(cur)
        self._preload()
        return batch
