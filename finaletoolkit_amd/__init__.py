"""
finaletoolkit_amd -- MI355X-native engine for FinaleToolkit's per-window hot
path (coverage, WPS, fragment-length histograms, DELFI short/long bins).

Host Python keeps the ``finaletoolkit.frag.*`` surface
(``finaletoolkit_amd.frag``); the per-fragment loops run as HIP kernels behind
the C ABI of ``include/ftk.h`` (``libftk_hip.so``).  No CPU fallback.
"""
import os as _os

# The streaming decoder keeps up to eight pieces of a file in flight on HIP streams of their own (inflate kernels of two or
# three pieces side by side, the row parser behind them on another stream); the HIP runtime multiplexes all streams of
# a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue run strictly one after the
# other.  Measured on the whole-genome DELFI leg (tools/e2e_genome_bench.py): 4 queues 0.176-0.183 s, 8: 0.161-0.181 s,
# 16: 0.151-0.162 s.  Read when the runtime initialises (first HIP call), so it is set here, before any; an explicit
# setting of the user's wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .exceptions import (FinaleToolkitError, InvalidInputError, MissingIndexError,  # noqa: F401
                         UnsupportedFormatError)

__version__ = "0.1.0"
