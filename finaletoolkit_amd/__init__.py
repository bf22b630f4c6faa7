"""
finaletoolkit_amd -- MI355X-native engine for FinaleToolkit's per-window hot
path (coverage, WPS, fragment-length histograms, DELFI short/long bins).

Host Python keeps the ``finaletoolkit.frag.*`` surface
(``finaletoolkit_amd.frag``); the per-fragment loops run as HIP kernels behind
the C ABI of ``include/ftk.h`` (``libftk_hip.so``).  No CPU fallback.
"""
from .exceptions import (FinaleToolkitError, InvalidInputError, MissingIndexError,  # noqa: F401
                         UnsupportedFormatError)

__version__ = "0.1.0"
