"""
Exception types raised on the hot path.  Names and base classes mirror the
reference (``src/finaletoolkit/exceptions.py:23-61``) so ``except ValueError``
/ ``except FileNotFoundError`` handlers keep working.
"""
from __future__ import annotations


class FinaleToolkitError(Exception):
    """Base class for all toolkit-specific errors."""


class InvalidInputError(FinaleToolkitError, ValueError):
    """Malformed or inconsistent user input."""


class UnsupportedFormatError(InvalidInputError):
    """Input file format the engine cannot read."""


class MissingReferenceError(InvalidInputError):
    """A reference genome is required but was not given."""


class MissingIndexError(FinaleToolkitError, FileNotFoundError):
    """A required index (.bai/.tbi) is missing."""


class ContigNotFoundError(InvalidInputError):
    """Requested contig absent from the input."""


__all__ = ["FinaleToolkitError", "InvalidInputError", "UnsupportedFormatError", "MissingReferenceError",
           "MissingIndexError", "ContigNotFoundError"]
