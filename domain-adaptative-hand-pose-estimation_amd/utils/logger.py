"""Run directory manager with the reference's layout (``utils/logger.py``): console output is mirrored into
``<root>/<phase>-<timestamp>.txt``; checkpoints live in ``<root>/checkpoints/<name>.pth`` and debug images in
``<root>/visualize/<epoch or phase>/``."""
import os
import sys
import time


class TextLogger:
    """A writable stream that forwards to the console stream and appends to a log file."""

    def __init__(self, filename, stream=None):
        self.terminal = stream if stream is not None else sys.stdout
        self.log = open(filename, 'a')

    def write(self, message):
        for sink in (self.terminal, self.log):
            sink.write(message)
        self.flush()

    def flush(self):
        for sink in (self.terminal, self.log):
            sink.flush()

    def close(self):
        self.log.close()


class _Null:
    def write(self, message):
        pass

    def flush(self):
        pass

    def close(self):
        pass


class CompleteLogger:
    """``quiet=True`` (additive; ranks > 0 of a data-parallel run): same directory layout and paths, but ordinary output
    is dropped instead of being mirrored -- rank 0 owns the console and the log file; stderr stays attached."""

    def __init__(self, root, phase='train', quiet=False):
        self.root, self.phase, self.epoch = root, phase, 0
        self.visualize_directory = os.path.join(root, 'visualize')
        self.checkpoint_directory = os.path.join(root, 'checkpoints')
        for d in (self.visualize_directory, self.checkpoint_directory):
            os.makedirs(d, exist_ok=True)
        self._saved = (sys.stdout, sys.stderr)
        if quiet:
            self.logger = _Null()
            sys.stdout = self.logger
            if phase != 'train':
                self.set_epoch(phase)
            return
        stamp = time.strftime('%Y-%m-%d-%H_%M_%S')
        path = os.path.join(root, '%s-%s.txt' % (phase, stamp))
        if os.path.exists(path):          # two runs within the same second: start the file afresh
            os.remove(path)
        self._saved = (sys.stdout, sys.stderr)
        self.logger = TextLogger(path, sys.stdout)
        sys.stdout = sys.stderr = self.logger
        if phase != 'train':
            self.set_epoch(phase)

    # the sub-directory images go to: the epoch number while training, the phase name otherwise
    def _bucket(self):
        return str(self.epoch if self.phase == 'train' else self.phase)

    def set_epoch(self, epoch):
        self.epoch = epoch
        os.makedirs(os.path.join(self.visualize_directory, str(epoch)), exist_ok=True)

    def get_image_path(self, filename: str):
        return os.path.join(self.visualize_directory, self._bucket(), filename)

    def get_checkpoint_path(self, name=None):
        return os.path.join(self.checkpoint_directory, '%s.pth' % (self._bucket() if name is None else name))

    def close(self):
        sys.stdout, sys.stderr = self._saved
        self.logger.close()
