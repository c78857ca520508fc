"""Small host utilities used by the entrypoint (the subset of /root/reference/ddpo/utils the DDPO path touches)."""
import logging
import os
import time

from . import prng  # noqa: F401
from .parser import Parser  # noqa: F401


class Timer:
    """Wall-clock stopwatch: calling it returns the seconds since the last reset (reference ddpo/utils/timer.py:4-13)."""

    def __init__(self):
        self._t0 = time.time()

    def __call__(self, reset=True):
        now = time.time()
        dt = now - self._t0
        if reset:
            self._t0 = now
        return dt


class fs:
    @staticmethod
    def join_and_create(*parts):
        """os.path.join + make the parent directory (reference ddpo/utils/filesystem.py:100-105)."""
        path = os.path.join(*parts)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        return path


def init_logging(name, verbose=False):
    """Quiet the chatty dependencies unless --verbose (reference ddpo/utils/logger.py:7-29)."""
    level = logging.INFO if verbose else logging.ERROR
    for dep in ("transformers", "urllib3", "PIL", "matplotlib"):
        logging.getLogger(dep).setLevel(level)
    logging.basicConfig(level=logging.INFO if verbose else logging.WARNING)
    return logging.getLogger(name)


def shard(x, n_devices=1):
    """Reshape the leading dim to (n_devices, -1, ...) (reference ddpo/utils/preprocessing.py:35-49); one process drives
    one GPU here, so n_devices is 1 and this only adds the reference's leading device axis."""
    return x.reshape(n_devices, -1, *x.shape[1:])


def unshard(x):
    return x.reshape(-1, *x.shape[2:])
