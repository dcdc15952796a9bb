"""File logger with the surface the DV solver path uses (reference utils/logger.py:5-52):
``Logging(log_path, experiment_name=None, source_file=None)``, ``.print(*args)``,
``.get_output_dir()``; each instance writes ``<log_path>/<timestamp>[_<name>]/output.log``."""
from __future__ import annotations

import datetime as _dt
import logging
import os

import numpy as np

_FLOATS = (float, np.floating)


class Logging:
    def __init__(self, log_path, experiment_name=None, source_file=None):
        self.log_path = log_path
        self.experiment_name = experiment_name
        stamp = _dt.datetime.now().strftime("%Y-%m-%d_%H-%M-%S-%f")
        if experiment_name is not None:
            stamp = f"{stamp}_{experiment_name}"
        self.output_dir = os.path.join(log_path, stamp)
        os.makedirs(self.output_dir, exist_ok=True)
        # one logger per output directory, so two runs in one process do not write into each other
        self.logger = logging.getLogger(f"qcpinn.{self.output_dir}")
        self.logger.setLevel(logging.DEBUG)
        self.logger.propagate = False
        self._handler = logging.FileHandler(os.path.join(self.output_dir, "output.log"), mode="w")
        self.logger.addHandler(self._handler)

    def get_output_dir(self):
        return self.output_dir

    @staticmethod
    def _fmt(v):
        return "%.4e" % v if isinstance(v, _FLOATS) else str(v)

    def print(self, *args):
        """All arguments on one line (floats as %.4e), like the reference's terminator juggling."""
        self.logger.info("".join(self._fmt(a) for a in args))
        self._handler.flush()
