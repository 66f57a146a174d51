"""Helpers the method classes and the task-batch loops need, under the reference's names
(reference: src/utils.py).  Only what the EM-Dirichlet path touches is provided; CLIP feature
extraction, datasets and the YAML merge are out of scope."""
import logging

import numpy as np


def compute_confidence_interval(data, axis=0):
    """Mean and 95 % confidence half-width (reference: src/utils.py:27-37)."""
    a = 1.0 * np.array(data)
    m = np.mean(a, axis=axis)
    std = np.std(a, axis=axis)
    pm = 1.96 * (std / np.sqrt(a.shape[axis]))
    return m, pm


class Logger:
    """File + stderr logger with per-instance handlers, removed by del_logger()
    (reference: src/utils.py:171-221).  log_file=None logs to stderr only."""

    def __init__(self, name, log_file=None, level=logging.INFO):
        self.logger = logging.getLogger(f"{name}.{id(self)}")
        self.logger.setLevel(level)
        self.logger.propagate = False
        self._handlers = []
        fmt = logging.Formatter("[%(name)s]: [%(levelname)s]: %(message)s")
        targets = [logging.StreamHandler()]
        if log_file:
            try:
                targets.append(logging.FileHandler(log_file))
            except OSError:
                pass
        for h in targets:
            h.setFormatter(fmt)
            self.logger.addHandler(h)
            self._handlers.append(h)

    def info(self, msg):
        self.logger.info(msg)

    def warning(self, msg):
        self.logger.warning(msg)

    def del_logger(self):
        for h in self._handlers:
            self.logger.removeHandler(h)
            try:
                h.close()
            except Exception:
                pass
        self._handlers = []


class CfgNode(dict):
    """Flat attribute dict, the shape of the reference's merged config (src/utils.py:40-88)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value
