"""`from climate_learn.dist.profile import *` must resolve (examples/intermediate_downscaling.py:42).  The
reference wraps gptl4py and every call site is commented out; here the timer is a HIP-event stopwatch
(rocprofv3 is the real profiler, see profiles/)."""
import time

import torch

__all__ = ["ProfileTimer"]


class ProfileTimer:
    def __init__(self):
        self._t = {}
        self.totals = {}

    def begin(self, name):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._t[name] = time.perf_counter()

    def end(self, name):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - self._t.pop(name)
        self.totals[name] = self.totals.get(name, 0.0) + dt
        return dt
