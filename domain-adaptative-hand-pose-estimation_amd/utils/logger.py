"""stdout/stderr tee + checkpoint / image paths (reference ``utils/logger.py``; same directory layout:
``<root>/<phase>-<time>.txt``, ``<root>/checkpoints/<name>.pth``, ``<root>/visualize/<epoch|phase>/``)."""
import os
import sys
import time


class TextLogger(object):
    """Writes stream output to an external text file as well."""

    def __init__(self, filename, stream=sys.stdout):
        self.terminal = stream
        self.log = open(filename, 'a')

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)
        self.flush()

    def flush(self):
        self.terminal.flush()
        self.log.flush()

    def close(self):
        self.log.close()


class CompleteLogger:
    def __init__(self, root, phase='train'):
        self.root, self.phase, self.epoch = root, phase, 0
        self.visualize_directory = os.path.join(root, "visualize")
        self.checkpoint_directory = os.path.join(root, "checkpoints")
        for d in (root, self.visualize_directory, self.checkpoint_directory):
            os.makedirs(d, exist_ok=True)
        now = time.strftime("%Y-%m-%d-%H_%M_%S", time.localtime(time.time()))
        log_filename = os.path.join(root, "{}-{}.txt".format(phase, now))
        if os.path.exists(log_filename):
            os.remove(log_filename)
        self._stdout, self._stderr = sys.stdout, sys.stderr
        self.logger = TextLogger(log_filename, sys.stdout)
        sys.stdout = self.logger
        sys.stderr = self.logger
        if phase != 'train':
            self.set_epoch(phase)

    def set_epoch(self, epoch):
        os.makedirs(os.path.join(self.visualize_directory, str(epoch)), exist_ok=True)
        self.epoch = epoch

    def _get_phase_or_epoch(self):
        return str(self.epoch) if self.phase == 'train' else self.phase

    def get_image_path(self, filename: str):
        return os.path.join(self.visualize_directory, self._get_phase_or_epoch(), filename)

    def get_checkpoint_path(self, name=None):
        if name is None:
            name = self._get_phase_or_epoch()
        return os.path.join(self.checkpoint_directory, str(name) + ".pth")

    def close(self):
        sys.stdout, sys.stderr = self._stdout, self._stderr
        self.logger.close()
