"""
TEST INFRASTRUCTURE ONLY.  Golden vectors for the host glue of ``multi_wps`` and ``ContigGaps``: runs the IMPORTED
reference (``/root/reference``, through oracle/refstub.py) in the build container on seeded random inputs and records
inputs + outputs as data in ``tests/golden/site_windows.json``:

* ``_read_sites`` (frag/_multi_wps.py:240-297): site BED text, interval size, contig lengths -> windows, warning texts
  (or the error text);
* ``ContigGaps.in_tcmere`` / ``get_arm`` (genome/gaps.py:217-267): intervals -> answers.

usage: python oracle/gen_golden_sites.py   (rewrites the fixture; tests/test_host_logic.py replays it)
"""
import json
import os
import random
import sys
import tempfile
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refstub  # noqa: E402

refstub.install()
from finaletoolkit.frag import _multi_wps as R  # noqa: E402
from finaletoolkit.genome.gaps import ContigGaps  # noqa: E402

rng = random.Random(20261003)
lengths = {"c1": 100000, "c2": 3000, "chr3": 50000}
tmp = tempfile.mkdtemp()
cases = []
for trial in range(48):
    lines = []
    for _ in range(rng.randint(0, 30)):
        c = rng.choice(["c1", "c1", "c1", "c2", "chr3", "zz"])
        a = rng.randint(0, 60000)
        b = a + rng.randint(0, 500)
        if trial % 12 == 11 and rng.random() < 0.1:
            a, b = b + 1, a  # start behind stop: the reference raises
        lines.append(f"{c}\t{a}\t{b}\n")
    if rng.random() < 0.5:
        lines.sort(key=lambda ln: (ln.split()[0], int(ln.split()[1])))
    text = "".join(lines)
    path = os.path.join(tmp, "s.bed")
    open(path, "w").write(text)
    size = rng.choice([2, 10, 1000, 5000, 40000])
    case = dict(bed=text, interval_size=size)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        try:
            contigs, starts, stops = R._read_sites(path, size, list(lengths), lengths)
            case.update(contigs=list(contigs), starts=[int(x) for x in starts], stops=[int(x) for x in stops])
        except ValueError as exc:
            case["error"] = str(exc).replace(path, "{path}")
    case["warnings"] = [str(w.message) for w in seen]
    cases.append(case)
gaps = []
for trial in range(60):
    cen = (rng.randint(0, 1000), rng.randint(0, 1000))
    tel = [(rng.randint(0, 1000), rng.randint(0, 1000)) for _ in range(rng.randint(0, 3))]
    short_arm = rng.random() < 0.5
    contig = rng.choice(["chr1", "13", "chrX"])
    g = ContigGaps(contig, cen, tel, short_arm)
    qs = []
    for _ in range(12):
        s, e = rng.randint(0, 1000), rng.randint(0, 1000)
        try:
            arm = g.get_arm(s, e)
        except ValueError as exc:
            arm = "ValueError: " + str(exc)
        qs.append([s, e, bool(g.in_tcmere(s, e)), arm])
    gaps.append(dict(contig=contig, centromere=list(cen), telomeres=[list(t) for t in tel], has_short_arm=short_arm, queries=qs))
out = os.path.join(ROOT, "tests", "golden", "site_windows.json")
json.dump(dict(lengths=lengths, sites=cases, gaps=gaps), open(out, "w"), indent=0)
print("wrote", out, len(cases), "site cases,", sum("error" in c for c in cases), "with errors;", len(gaps), "gap cases")
