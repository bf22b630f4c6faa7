"""Where a command's wall time went: a lap clock whose laps a module publishes as ``LAST_STAGE_S`` (read by bench.py /
tools/cmd_legs.py; ``frag/_delfi.py`` keeps its own, older form)."""
from __future__ import annotations

import time


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
