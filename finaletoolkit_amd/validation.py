"""
Contig and interval checks of the reference's utility layer (``utils/validation.py:14-183``: ``validate_compatible_contigs``
guards the CLI's reference / input pairing, ``valid_interval`` the DELFI bins, ``frag/_delfi.py:476-478``), kept with
the reference's arguments, return values, exception types and messages.  Pure Python: nothing here touches the GPU.

Both functions work the same way: the first problem found is either raised (its exception type is part of the
contract) or logged and answered with ``False``.
"""
from __future__ import annotations

import logging

__all__ = ["validate_compatible_contigs", "valid_interval"]

logger = logging.getLogger(__name__)


def _settle(problem, throw_on_error: bool) -> bool:
    """``problem``: None (valid) or ``(exception type, message)``."""
    if problem is None:
        return True
    kind, msg = problem
    if throw_on_error:
        raise kind(msg)
    logger.error(msg)
    return False


def _contig_problem(reference_contigs, input_contigs, allow_subset, validate_sizes):
    ref, inp = set(reference_contigs), set(input_contigs)  # (a dict gives its keys)
    if inp - ref:
        return ValueError, f"Input contains contigs not found in reference: {sorted(inp - ref)}"
    if not allow_subset and ref - inp:
        return ValueError, f"Reference contains contigs not found in input: {sorted(ref - inp)}"
    if not validate_sizes:
        return None
    if not (isinstance(reference_contigs, dict) and isinstance(input_contigs, dict)):
        return TypeError, ("validate_sizes=True requires both reference_contigs and input_contigs to be dictionaries "
                           "with lengths.")
    for contig in inp:  # (only what the input holds: the reference may know more contigs)
        if reference_contigs[contig] != input_contigs[contig]:
            return RuntimeError, (f"Contig length mismatch for '{contig}': reference={reference_contigs[contig]}, "
                                  f"input={input_contigs[contig]}")
    return None


def validate_compatible_contigs(reference_contigs, input_contigs, allow_subset: bool = True,
                                validate_sizes: bool = False, throw_on_error: bool = True) -> bool:
    """Are the input's contigs those of the reference (names; lengths too with ``validate_sizes``, which needs two
    dicts)?  ``allow_subset=False`` asks for the same set on both sides.  Raises ``ValueError`` (names), ``TypeError``
    (sizes asked of lists) or ``RuntimeError`` (lengths differ) unless ``throw_on_error`` is false."""
    return _settle(_contig_problem(reference_contigs, input_contigs, allow_subset, validate_sizes), throw_on_error)


def _interval_problem(reference_contigs, contig, start, stop):
    if contig not in reference_contigs:
        return ValueError, f"Contig '{contig}' not found in reference."
    if not isinstance(reference_contigs, dict):  # names only: nothing to hold the interval against but zero
        if start is not None and start < 0:
            return IndexError, f"Start position {start} cannot be negative."
        return None
    length = reference_contigs[contig]
    for what, pos, ok in (("Start", start, lambda v: 0 <= v < length), ("Stop", stop, lambda v: 0 <= v <= length)):
        if pos is not None and not ok(pos):
            return IndexError, f"{what} position {pos} is out of bounds for contig '{contig}' (length {length})."
    if start is not None and stop is not None and start >= stop:
        return ValueError, f"Invalid interval: start ({start}) must be less than stop ({stop})."
    return None


def valid_interval(reference_contigs, contig: str, start: int | None = None, stop: int | None = None,
                   throw_on_error: bool = False) -> bool:
    """Is ``contig:[start, stop)`` an interval of the reference (0-based start, stop up to the contig's length)?  With
    a list of names only the contig and a negative start can be wrong."""
    return _settle(_interval_problem(reference_contigs, contig, start, stop), throw_on_error)
