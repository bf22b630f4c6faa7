"""Genome annotations used by DELFI (centromere / telomere / short-arm gap tracks)."""
from .gaps import ContigGaps, GenomeGaps  # noqa: F401
