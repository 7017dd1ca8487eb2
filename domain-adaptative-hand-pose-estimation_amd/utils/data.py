"""Endless batch source over a finite loader (API of the reference's ``utils/data.py``:
``ForeverDataIterator(loader)`` with ``next()`` and ``len()``)."""
import itertools


def _cycle_fresh(loader):
    # itertools.cycle would replay the cached first epoch; a shuffling loader must be re-iterated instead
    while True:
        empty = True
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
