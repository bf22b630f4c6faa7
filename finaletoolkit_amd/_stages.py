"""Where a command's wall time went: a lap clock whose laps a module publishes as ``LAST_STAGE_S`` (read by bench.py /
tools/cmd_legs.py; ``frag/_delfi.py`` keeps its own, older form)."""
from __future__ import annotations

import contextlib
import functools
import gc
import time


@contextlib.contextmanager
def collector_paused():
    """The cyclic collector off for the span of a command that builds tens of thousands of small result objects (rows,
    named tuples, interval tuples): none of them is garbage, but every generation-2 pass walks all of them - a 70 ms
    stall between two contigs of a 70 ms command when it strikes (seen in ``frag_length_intervals``)."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def without_collector(fn):
    """Decorator form of ``collector_paused`` for a command's entry point."""
    @functools.wraps(fn)
    def run(*args, **kwargs):
        with collector_paused():
            return fn(*args, **kwargs)
    return run


class Stages(dict):
    def __init__(self):
        super().__init__()
        self._t0 = self._t = time.perf_counter()

    def lap(self, name: str) -> None:
        now = time.perf_counter()
        self[name] = self.get(name, 0.0) + now - self._t
        self._t = now

    def publish(self, target: dict) -> None:
        self.lap("other")
        target.clear()
        target.update({k: round(v, 4) for k, v in self.items()})
        target["total"] = round(time.perf_counter() - self._t0, 4)
