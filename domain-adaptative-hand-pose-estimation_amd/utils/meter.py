"""Console meters with the reference's public names and printed format (``utils/meter.py``):
``AverageMeter('Loss', ':.2e')`` prints ``Loss 1.23e+00 (1.20e+00)``, ``ProgressMeter`` joins meters with tabs
behind ``prefix[  i/500]``."""
from typing import Iterable, Optional


def _fmt(spec: str, value) -> str:
    return ('{' + spec + '}').format(value)


class Meter:
    """Holds the latest value only."""

    def __init__(self, name: str, fmt: Optional[str] = ':f'):
        self.name, self.fmt = name, fmt
        self.val = 0

    def reset(self):
        self.val = 0

    def update(self, val):
        self.val = val

    def __str__(self):
        return '%s %s' % (self.name, _fmt(self.fmt, self.val))


class AverageMeter(Meter):
    """Latest value plus the weighted running mean (weights = sample counts)."""

    def __init__(self, name: str, fmt: Optional[str] = ':f'):
        super().__init__(name, fmt)
        self.sum = self.count = self.avg = 0

    def reset(self):
        super().reset()
        self.sum = self.count = self.avg = 0

    def update(self, val, n=1):
        self.val = val
        self.sum, self.count = self.sum + val * n, self.count + n
        if self.count > 0:
            self.avg = self.sum / self.count

    def output(self):
        return self.avg

    def __str__(self):
        return '%s (%s)' % (super().__str__(), _fmt(self.fmt, self.avg))


class AverageMeterDict:
    """One AverageMeter per key (per key-point group in validate())."""

    def __init__(self, names: Iterable[str], fmt: Optional[str] = ':f'):
        self.dict = {n: AverageMeter(n, fmt) for n in names}

    def __getitem__(self, key):
        return self.dict[key]

    def reset(self):
        for m in self.dict.values():
            m.reset()

    def update(self, values, n=1):
        for key, v in values.items():
            self.dict[key].update(v, n)

    def average(self):
        return {key: m.avg for key, m in self.dict.items()}


class ProgressMeter:
    def __init__(self, num_batches, meters, prefix=""):
        self.meters, self.prefix = list(meters), prefix
        self._width = len(str(int(num_batches)))
        self._total = ('{:%dd}' % self._width).format(num_batches)

    def display(self, batch):
        head = '%s[%s/%s]' % (self.prefix, ('{:%dd}' % self._width).format(batch), self._total)
        print('\t'.join([head] + [str(m) for m in self.meters]))
