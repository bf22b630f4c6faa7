"""
UCSC gap-track lookups used by DELFI: which bins overlap an assembly gap, which
chromosome arm a bin is on, and the per-contig centromere / telomere constants
the DELFI kernel tests every fragment against.

Same classes and semantics as the reference's ``genome/gaps.py:40-267``
(``GenomeGaps``, ``ContigGaps.in_tcmere`` with its deliberate ``all()`` over
telomeres, ``get_arm``); the bundled tracks are the UCSC hg19 / hg38 ``gap``
tables reduced to BED4 (``genome/data/*.gaps.bed.gz``).
"""
from __future__ import annotations

import gzip
import os
from typing import Iterable, Optional, Union

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_GAP_DTYPE = [("contig", "<U32"), ("start", "<i8"), ("stop", "<i8"), ("type", "<U32")]

__all__ = ["GenomeGaps", "ContigGaps"]


def _read_bed4(path) -> np.ndarray:
    opener = gzip.open if str(path).endswith(".gz") else open
    rows = []
    with opener(path, "rt") as fh:
        for line in fh:
            if not line.strip() or line.startswith("#"):
                continue
            p = line.split()
            rows.append((p[0], int(p[1]), int(p[2]), p[3]))
    return np.array(rows, dtype=_GAP_DTYPE)


class GenomeGaps:
    """Telomere / centromere / short-arm intervals of a reference genome."""

    def __init__(self, gaps_bed: Union[os.PathLike, str, None] = None) -> None:
        if gaps_bed is None:
            return
        if isinstance(gaps_bed, str) and gaps_bed in ("hg19", "b37", "human_g1k_v37", "hg38", "GRCh38") \
                and not os.path.exists(gaps_bed):
            named = {"hg19": GenomeGaps.ucsc_hg19, "b37": GenomeGaps.b37, "human_g1k_v37": GenomeGaps.b37,
                     "hg38": GenomeGaps.hg38, "GRCh38": GenomeGaps.hg38}[gaps_bed]()
            self._set_gaps(named.gaps)
            return
        self._set_gaps(_read_bed4(gaps_bed))

    def _set_gaps(self, gaps: np.ndarray) -> None:
        self.centromeres = gaps[gaps["type"] == "centromere"]
        self.telomeres = gaps[gaps["type"] == "telomere"]
        self.short_arms = gaps[gaps["type"] == "short_arm"]
        self.gaps = gaps

    @classmethod
    def _from_track(cls, name: str, strip_chr: bool = False) -> "GenomeGaps":
        g = cls()
        gaps = _read_bed4(os.path.join(_DATA, f"{name}.gaps.bed.gz"))
        if strip_chr:
            gaps["contig"] = np.char.replace(gaps["contig"], "chr", "")
        g._set_gaps(gaps)
        return g

    @classmethod
    def ucsc_hg19(cls) -> "GenomeGaps":
        return cls._from_track("hg19")

    @classmethod
    def b37(cls) -> "GenomeGaps":
        return cls._from_track("hg19", strip_chr=True)

    @classmethod
    def hg38(cls) -> "GenomeGaps":
        return cls._from_track("hg38")

    def get_contig_gaps(self, contig: str) -> Optional["ContigGaps"]:
        """genome/gaps.py:157-169: ``None`` when the contig has no centromere row."""
        cen = self.centromeres[self.centromeres["contig"] == contig]
        if cen.shape[0] == 0:
            return None
        tel = self.telomeres[self.telomeres["contig"] == contig]
        short_arm = self.short_arms[self.short_arms["contig"] == contig]
        return ContigGaps(contig, (int(cen[0]["start"]), int(cen[0]["stop"])),
                          [(int(t["start"]), int(t["stop"])) for t in tel], short_arm.shape[0] > 0)

    def to_bed(self, output_file) -> None:
        gaps = np.sort(self.gaps)
        lines = "".join(f"{g['contig']}\t{g['start']}\t{g['stop']}\t{g['type']}\n" for g in gaps)
        if str(output_file).endswith(".gz"):
            with gzip.open(output_file, "wt") as out:
                out.write(lines)
        elif str(output_file) == "-":
            import sys
            sys.stdout.write(lines)
        else:
            with open(output_file, "w") as out:
                out.write(lines)


class ContigGaps:
    """Centromere / telomere intervals of one contig (the reference's ``genome/gaps.py:202-267``), kept in the form
    the DELFI kernel tests fragments against (``as_kernel_constants`` -> ``ftk_gaps``): one centromere interval, and
    ONE interval standing for "overlaps every telomere" - ``all_i(stop > a_i and start < b_i)`` is
    ``stop > max a_i and start < min b_i`` (the reference's deliberate ``all()``: its DELFI outputs were produced
    with it).  The Python predicates below are evaluated on those constants, so host and kernel cannot disagree."""

    def __init__(self, contig: str, centromere: tuple[int, int], telomeres: Iterable[tuple[int, int]],
                 has_short_arm: bool = False) -> None:
        self.contig = contig
        self.centromere = centromere
        self.telomeres = list(telomeres)
        self.has_short_arm = has_short_arm

    def _every_telomere(self) -> tuple[float, float]:
        """The interval a fragment must overlap to overlap every telomere (nothing does when there are none)."""
        if not self.telomeres:
            return float("inf"), float("-inf")
        return max(t[0] for t in self.telomeres), min(t[1] for t in self.telomeres)

    def in_tcmere(self, start: int, stop: int) -> bool:
        lo, hi = self._every_telomere()
        return any(stop > a and start < b for a, b in (self.centromere, (lo, hi)))

    def get_arm(self, start: int, stop: int) -> str:
        if stop < start:
            raise ValueError("start must be less than stop")
        left_of, right_of = stop < self.centromere[0], start > self.centromere[1]
        if (left_of and self.has_short_arm) or not (left_of or right_of):
            return "NOARM"  # acrocentric short arm, or touching the centromere
        return self.contig.replace("chr", "") + ("p" if left_of else "q")

    def as_kernel_constants(self):
        """(cen_start, cen_stop, [(t0, t1), ...]) for ``ftk_gaps``."""
        return (self.centromere[0], self.centromere[1], list(self.telomeres))


def ucsc_hg19_gap_bed(output_file) -> None:
    """BED4 of the UCSC hg19 centromeres / telomeres / short arms (genome/gaps.py:270-272)."""
    return GenomeGaps.ucsc_hg19().to_bed(output_file)


def b37_gap_bed(output_file) -> None:
    """The same for Broad b37 / human_g1k_v37 contig names (genome/gaps.py:275-280)."""
    return GenomeGaps.b37().to_bed(output_file)


def ucsc_hg38_gap_bed(output_file) -> None:
    """The same for UCSC hg38 (genome/gaps.py:283-285)."""
    return GenomeGaps.hg38().to_bed(output_file)


def _cli_gap_bed(reference_genome: str, output_file: str) -> None:
    """CLI ``gap-bed`` (genome/gaps.py:288-302)."""
    tracks = {"hg19": ucsc_hg19_gap_bed, "b37": b37_gap_bed, "human_g1k_v37": b37_gap_bed, "hg38": ucsc_hg38_gap_bed,
              "GRCh38": ucsc_hg38_gap_bed}
    if reference_genome not in tracks:
        raise ValueError(f"Gap track for {reference_genome} is currently unavailable. It is possible to create a gap "
                         "track de novo if interval data for centromeres, telomeres, and short_arms exist for the "
                         "reference sequence of interest.")
    tracks[reference_genome](output_file)
