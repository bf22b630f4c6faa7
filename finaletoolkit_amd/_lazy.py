"""Modules that are imported when first used.  A command-line call is a fresh process whose wall time is mostly
start-up (``tools/cold_start_probe.py``): pandas alone is 0.2-0.6 s of it, and only the DELFI frame needs it."""
from __future__ import annotations

import importlib


class LazyModule:
    """Stands where ``import <name>`` stood; the import happens at the first attribute access."""

    def __init__(self, name: str):
        self.__dict__["_name"] = name
        self.__dict__["_mod"] = None

    def _load(self):
        if self.__dict__["_mod"] is None:
            self.__dict__["_mod"] = importlib.import_module(self.__dict__["_name"])
        return self.__dict__["_mod"]

    def __getattr__(self, attr):
        return getattr(self._load(), attr)

    def __dir__(self):
        return dir(self._load())

    def __repr__(self):
        return f"<lazy module {self.__dict__['_name']!r}{'' if self.__dict__['_mod'] is None else ' (loaded)'}>"
