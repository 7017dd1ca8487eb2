"""Running meters and the progress line (reference ``utils/meter.py``; same printed format)."""
from typing import List, Optional


class AverageMeter(object):
    """Computes and stores the average and current value."""

    def __init__(self, name: str, fmt: Optional[str] = ':f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        if self.count > 0:
            self.avg = self.sum / self.count

    def __str__(self):
        return ('{name} {val' + self.fmt + '} ({avg' + self.fmt + '})').format(**self.__dict__)

    def output(self):
        return self.avg


class AverageMeterDict(object):
    def __init__(self, names: List, fmt: Optional[str] = ':f'):
        self.dict = {name: AverageMeter(name, fmt) for name in names}

    def reset(self):
        for m in self.dict.values():
            m.reset()

    def update(self, accuracies, n=1):
        for name, acc in accuracies.items():
            self.dict[name].update(acc, n)

    def average(self):
        return {name: m.avg for name, m in self.dict.items()}

    def __getitem__(self, item):
        return self.dict[item]


class Meter(object):
    """Computes and stores the current value."""

    def __init__(self, name: str, fmt: Optional[str] = ':f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = 0

    def update(self, val):
        self.val = val

    def __str__(self):
        return ('{name} {val' + self.fmt + '}').format(**self.__dict__)


class ProgressMeter(object):
    def __init__(self, num_batches, meters, prefix=""):
        width = len(str(num_batches // 1))
        self.batch_fmtstr = '[{:' + str(width) + 'd}/' + ('{:' + str(width) + 'd}').format(num_batches) + ']'
        self.meters, self.prefix = meters, prefix

    def display(self, batch):
        print('\t'.join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters]))
