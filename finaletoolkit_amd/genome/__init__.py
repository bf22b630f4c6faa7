"""Genome annotations used by DELFI (centromere / telomere / short-arm gap tracks)."""
from .gaps import (ContigGaps, GenomeGaps, b37_gap_bed, ucsc_hg19_gap_bed,  # noqa: F401
                   ucsc_hg38_gap_bed)
