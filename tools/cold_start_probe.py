"""Where a FRESH process spends its time before the first result: interpreter + imports, library load, context
(HIP runtime start, code objects, streams), first decode of a small file, first kernel, and the same calls again
(warm).  One CLI invocation = one such process, so these are the costs a command-line user pays on every call.

    python tools/cold_start_probe.py            # spawns itself N times, prints min / median per stage
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FRAG = os.path.join(ROOT, "tests", "data", "12.3444.b37.frag.gz")
BED = os.path.join(ROOT, "tests", "data", "intervals.bed")


def child():
    t = [time.perf_counter()]
    marks = {}

    def mark(name):
        t.append(time.perf_counter())
        marks[name] = round(t[-1] - t[-2], 4)

    import numpy as np  # noqa: F401
    mark("import_numpy")
    sys.path.insert(0, ROOT)
    import finaletoolkit_amd  # noqa: F401
    mark("import_package")
    from finaletoolkit_amd import frag  # noqa: F401
    mark("import_frag")
    from finaletoolkit_amd import _lib, source
    _lib.load()
    mark("load_library")
    eng = source.get_engine()
    mark("context")
    eng.sync()
    mark("first_sync")
    src = source.open_source(FRAG)
    key = src.require("12")
    mark("first_decode")
    eng.window_counts(key, [34442500], [34446500])
    mark("first_kernel")
    eng.window_counts(key, [34442500], [34446500])
    mark("second_kernel")
    source.close_all()
    mark("close")
    t0 = time.perf_counter()
    r = frag.coverage(FRAG, BED, None)
    marks["frag_coverage_after_close"] = round(time.perf_counter() - t0, 4)
    marks["rows"] = len(r)
    print(json.dumps(marks))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    rows = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(out.stdout, out.stderr)
            raise SystemExit(1)
        d = json.loads(line[-1])
        d["process_wall"] = round(wall, 4)
        rows.append(d)
    keys = list(rows[0])
    print(f"{'stage':32s} {'min':>8s} {'median':>8s}")
    for k in keys:
        v = sorted(r[k] for r in rows)
        print(f"{k:32s} {v[0]:8.4f} {v[len(v) // 2]:8.4f}")
    # the command line itself
    cli = []
    for _ in range(n):
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", "coverage", FRAG, BED, "-o", os.devnull],
                       cwd=ROOT, capture_output=True)
        cli.append(time.perf_counter() - t0)
    cli.sort()
    print(f"{'cli coverage (whole process)':32s} {cli[0]:8.4f} {cli[len(cli) // 2]:8.4f}")


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
