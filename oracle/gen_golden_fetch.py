"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

Golden vectors for the ``AlignmentWrapper`` facade: the IMPORTED reference's class (io/alignment.py:74-302, over the
tabix stand-in of oracle/refstub.py) is opened on the reference's own fixtures and on tests/golden/synth.frag.gz and
asked for regions; the fragments it yields are recorded as data in tests/golden/fetch.json (regions at contig edges,
open bounds, empty regions, the whole file, three mapq cuts).  BAM input cannot be pinned this way (no pysam here):
tests check it against the fixture's 17 known fragments and against the C oracle.

Usage:  python oracle/gen_golden_fetch.py
"""
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import refstub  # noqa: E402

refstub.install()
from finaletoolkit.io.alignment import AlignmentWrapper  # noqa: E402

files = {"fixture": os.path.join("tests", "data", "12.3444.b37.frag.gz"),
         "fixture_bed6": os.path.join("tests", "data", "12.3444.b37.frag.bed.gz"),
         "synth": os.path.join("tests", "golden", "synth.frag.gz")}
regions = {
    "fixture": [("12", None, None), ("12", 34443118, 34443284), ("12", 34443284, 34443285), ("12", 34443283, 34443284),
                ("12", 34444000, 34446000), ("12", 0, 34443118), ("12", 34446652, 34446653), ("12", 34446653, None),
                ("12", None, 34443119), (None, None, None), (None, 5, 6)],
    "fixture_bed6": [("12", None, None), ("12", 34444000, 34446000), (None, None, None)],
    "synth": [("chrA", 0, 1), ("chrA", 100_000, 100_500), ("chrA", 399_000, 400_000), ("chrA", 250_000, 250_001),
              ("chrB", None, 3_000), ("chrB", 149_000, None), ("chrB", 70_000, 70_000), ("chrA", 163_840, 163_841),
              ("chrA", 16_383, 16_385)],
}
out = {}
for tag, rel in files.items():
    cases = []
    for q in (0, 30, 60):
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            with AlignmentWrapper(os.path.join(ROOT, rel), quality_threshold=q) as aw:
                chroms = dict(aw.chroms)
                for contig, a, b in regions[tag]:
                    rows = [[f.contig, f.start, f.stop, f.mapq, bool(f.is_forward)] for f in aw.fetch(contig, a, b)]
                    cases.append(dict(quality_threshold=q, contig=contig, start=a, stop=b, fragments=rows))
        texts = sorted({str(w.message) for w in seen})
    out[tag] = dict(path=rel, chroms=chroms, is_sam=False, warnings=texts, cases=cases)
path = os.path.join(ROOT, "tests", "golden", "fetch.json")
json.dump(out, open(path, "w"))
print("wrote", path, {k: (len(v["cases"]), sum(len(c["fragments"]) for c in v["cases"])) for k, v in out.items()})
