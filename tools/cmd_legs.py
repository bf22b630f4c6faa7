#!/usr/bin/env python3
"""
File -> result legs of the reference COMMANDS other than `delfi`, timed through the product functions
(SURVEY 8-d "end-to-end"; reference frag/_coverage.py:145-305, frag/_multi_wps.py:31-223,300-341,
frag/_frag_length.py:333-508,511-640):

    coverage                frag.coverage(genome.frag.gz, 30 970 x 100 kb BED, out.bed, normalize=True)
    multi_wps_bw            frag.multi_wps(genome.frag.gz, 20 000 sites +- 2.5 kb -> out.bw)
    multi_wps_bedgz         ... -> out.bed.gz
    frag_length_intervals   frag.frag_length_intervals(genome.frag.gz, the same BED, out.bed)
    frag_length_bins        frag.frag_length_bins(genome.frag.gz) genome-wide, TSV with summary statistics

Each leg: repetitions (first / median / best), the stage split the function reports (`LAST_STAGE_S` of its module),
`results_ok` against the oracle on one whole contig (checker only, untimed), a floor and the reference-shaped
single-thread Python rate of the same unit of work on a small sample.

`bench.py` imports `measure`; run alone (`python tools/cmd_legs.py [--scale 0.1]`) it writes its own genome file.
"""
from __future__ import annotations

import argparse
import gzip
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from finaletoolkit_amd import synth  # noqa: E402
from finaletoolkit_amd.synth import gen_contig_device  # noqa: E402

WINDOW, MAPQ = 100_000, 30
CHECK = "21"


def rep_summary(runs):
    times = [r["total_s"] for r in runs]
    out = dict(min(runs, key=lambda r: r["total_s"]))
    out.update(first_s=round(times[0], 4), median_s=round(float(np.median(times)), 4), best_s=round(min(times), 4),
               repetitions=len(times), all_s=[round(t, 4) for t in times])
    return out


def write_side_files(tmp, sizes, n_sites=20_000, seed=4711):
    """chrom.sizes, the 100 kb tiling as a BED4 and a site BED of `n_sites` one-base sites spread by contig length."""
    cs = os.path.join(tmp, "legs.chrom.sizes")
    bed = os.path.join(tmp, "legs_windows.bed")
    sites = os.path.join(tmp, "legs_sites.bed")
    with open(cs, "w") as fh:
        fh.write("".join(f"{c}\t{n}\n" for c, n in sizes.items()))
    n_win = 0
    with open(bed, "w") as fh:
        for c, n in sizes.items():
            ws, we = synth.tiling_windows(n, WINDOW)
            fh.write("".join(f"{c}\t{a}\t{b}\tw{n_win + k}\n" for k, (a, b) in enumerate(zip(ws.tolist(), we.tolist()))))
            n_win += len(ws)
    rng = np.random.default_rng(seed)
    total = float(sum(sizes.values()))
    rows = []
    for c, n in sizes.items():
        k = max(1, int(round(n_sites * n / total)))
        pos = np.sort(rng.integers(3_000, max(n - 3_000, 3_001), k))
        rows += [(c, int(p)) for p in pos]
    with open(sites, "w") as fh:
        fh.write("".join(f"{c}\t{p}\t{p + 1}\n" for c, p in rows))
    return cs, bed, sites, n_win, len(rows)


def contig_frags(torch, dev, sizes, names, c, depth):
    k = names.index(c)
    return tuple(t.cpu().numpy() for t in gen_contig_device(torch, dev, sizes[c], synth.n_fragments(sizes[c], depth), synth.SEED_BASE + k))


def measure(torch, tmp, genome_file, sizes, threads, raw_floor=None, depth=30.0, reps=3, n_sites=20_000):
    """`genome_file`: a frag.gz of `sizes`' contigs written with seed SEED_BASE + contig index (bench.py's write_genome)."""
    from finaletoolkit_amd import frag, source
    from finaletoolkit_amd.bigwig import BigWigFile
    from finaletoolkit_amd.frag import _coverage as FC, _frag_length as FL, _runs as FR
    from oracle import oracle as O
    dev = torch.device("cuda", torch.cuda.current_device())
    names = list(sizes)
    cs, bed, sites, n_win, n_site_rows = write_side_files(tmp, sizes, n_sites)
    s, e, q, st = contig_frags(torch, dev, sizes, names, CHECK, depth)
    fr = O.Frags(s, e, q, st)
    rows_check = None
    ws21, we21 = synth.tiling_windows(sizes[CHECK], WINDOW)
    total_cov = 0
    hist_all = np.zeros(1001, np.int64)
    for k, c in enumerate(names):  # the whole file's mapq >= 30 fragments and their length distribution (normalisation, bins)
        cs_, ce_, cq_, _ = gen_contig_device(torch, dev, sizes[c], synth.n_fragments(sizes[c], depth), synth.SEED_BASE + k)
        ok = cq_ >= MAPQ
        total_cov += int(ok.sum().item())
        hist_all += torch.bincount((ce_ - cs_)[ok].to(torch.int64), minlength=1001)[:1001].cpu().numpy()
        del cs_, ce_, cq_, ok
    res = {}

    def timed(call, stages=None, n=reps):
        runs = []
        for _ in range(n):
            source.close_all()
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = call()
            runs.append(dict(total_s=round(time.perf_counter() - t0, 4), stages_s=dict(stages() if stages else {})))
        return rep_summary(runs), out

    def floor(leg, extra_s=0.0, note=None):
        if raw_floor:
            f = dict(raw_floor)
            if extra_s:  # (the output is compressed beside the decode: the floor is the longer of the two)
                f["decode_floor_s"] = f["floor_s"]
                f["floor_s"] = round(max(f["floor_s"], extra_s), 4)
                f["output_term_s"] = round(extra_s, 4)
                f["output_term"] = note
            f["frac_of_floor_best"] = round(f["floor_s"] / leg["best_s"], 3)
            f["frac_of_floor_median"] = round(f["floor_s"] / leg["median_s"], 3)
            leg["floor"] = f

    def py_rate(fn, units, cap_s=6.0):
        t0 = time.perf_counter()
        n = 0
        for u in units:
            fn(u)
            n += 1
            if time.perf_counter() - t0 > cap_s:
                break
        return n / (time.perf_counter() - t0), n

    s64, e64 = s.astype(np.int64), e.astype(np.int64)

    def fetched(a, b):
        lo = int(np.searchsorted(s64, a - 1000, side="left"))
        hi = int(np.searchsorted(s64, b, side="left"))
        k = np.flatnonzero(e64[lo:hi] > a) + lo
        return list(zip(s[k].tolist(), e[k].tolist(), q[k].tolist(), st[k].tolist()))

    # ---------------------------------------------------------------- coverage(normalize=True) -> .bed
    out_bed = os.path.join(tmp, "legs_cov.bed")
    leg, got = timed(lambda: frag.coverage(genome_file, bed, out_bed, normalize=True, scale_factor=1e6, workers=threads),
                     lambda: getattr(FC, "LAST_STAGE_S", {}))
    want = O.c_window_counts(fr, ws21, we21, mapq_min=MAPQ)
    mine = [r for r in got if r[0] == CHECK]
    ok = len(got) == n_win and len(mine) == len(ws21) and all(r.coverage == int(w) * (1e6 / total_cov) for r, w in zip(mine, want))
    ok = ok and sum(1 for _ in open(out_bed)) == n_win
    rate, n = py_rate(lambda k: O.py_single_coverage(fetched(int(ws21[k]), int(we21[k])), int(ws21[k]), int(we21[k]), None, None, "midpoint", MAPQ),
                      range(len(ws21)))
    leg.update(windows=n_win, windows_per_s=round(n_win / leg["best_s"], 1), results_ok=bool(ok),
               checked=f"contig {CHECK}: {len(ws21)} rows == C oracle counts x 1e6 / {total_cov}; file rows",
               reference_shaped_python_windows_per_s=round(rate, 2), python_sample=f"{n} windows of contig {CHECK}",
               x_vs_reference_shaped_python=round(n_win / leg["best_s"] / rate, 1))
    floor(leg)
    res["coverage_normalize_bed"] = leg

    # ---------------------------------------------------------------- multi_wps -> .bw / .bed.gz
    site_rows = [ln.split() for ln in open(sites)]
    mids21 = [int(r[1]) for r in site_rows if r[0] == CHECK]
    bases = 0
    for c in names:
        m = np.array([int(r[1]) for r in site_rows if r[0] == c], np.int64)
        a, b = np.maximum(m - 2500, 0), np.minimum(m + 2500, sizes[c])
        if len(m) > 1:
            b[:-1] = np.minimum(b[:-1], a[1:])
        bases += int(np.maximum(b - a, 0).sum())
    # what the output side of these two legs cannot go below: their bytes through the container's compressor (zlib
    # streams of float32 sections; gzip members of text rows - both at level 6, libdeflate when the box has it) at the
    # rate ONE thread reaches on a 4 Mb sample of this very track, times the threads the writers use
    from finaletoolkit_amd import writers
    eng = source.get_engine()
    probe_key = source.open_source(genome_file).require(CHECK)
    a0 = sizes[CHECK] // 2
    n_probe = min(4 << 20, sizes[CHECK] - a0)
    sample = eng.wps(probe_key, a0, a0 + n_probe, sizes[CHECK])
    t0 = time.perf_counter()
    writers.bigwig_sections(0, [a0], sample, None, 16384, 6, 1)
    bw_1t = 4 * n_probe / (time.perf_counter() - t0)          # float32 bytes per second, one thread
    with writers.bedgraph_rows(CHECK, [a0], sample[:1 << 20], None, 1) as rows:
        text_per_base = rows.n / float(1 << 20)
        t0 = time.perf_counter()
        rows.gzip_bytes(writers.GZIP_LEVEL, 1)
        gz_1t = rows.n / (time.perf_counter() - t0)           # text bytes per second, one thread
    source.close_all()
    out_floor = {".bw": (4 * bases / (bw_1t * threads), f"{4 * bases / 1e6:.0f} MB of float32 sections through zlib level 6 at "
                         f"{bw_1t / 1e6:.0f} MB/s per thread (measured on a 4 Mb sample) x {threads} threads"),
                 ".bed.gz": (text_per_base * bases / (gz_1t * threads), f"{text_per_base * bases / 1e6:.0f} MB of rows through gzip level "
                             f"{writers.GZIP_LEVEL} at {gz_1t / 1e6:.0f} MB/s per thread (measured on 1 Mi rows) x {threads} threads")}
    for suffix, key in ((".bw", "multi_wps_bw"), (".bed.gz", "multi_wps_bedgz")):
        out = os.path.join(tmp, "legs_wps" + suffix)
        leg, _ = timed(lambda: frag.multi_wps(genome_file, sites, cs, out, interval_size=5000, workers=threads),
                       lambda: getattr(FR, "LAST_STAGE_S", {}), n=reps if suffix == ".bw" else 2)
        ok, n_chk = True, 0
        if suffix == ".bw":
            with BigWigFile(out) as bw:
                for m in mids21[:: max(1, len(mids21) // 12)]:
                    a, b = max(m - 2500, 0), min(m + 2500, sizes[CHECK])
                    nxt = [x for x in mids21 if x > m]
                    if nxt:
                        b = min(b, max(nxt[0] - 2500, 0))
                    if b <= a:
                        continue
                    ok = ok and np.array_equal(np.asarray(bw.values(CHECK, a, b), np.float64), O.c_wps(fr, a, b, sizes[CHECK]).astype(np.float64))
                    n_chk += 1
        else:
            m = mids21[0]
            a, b = max(m - 2500, 0), min(m + 2500, sizes[CHECK])
            if len(mids21) > 1:
                b = min(b, max(mids21[1] - 2500, 0))
            want = O.c_wps(fr, a, b, sizes[CHECK])
            vals, n_rows = [], 0
            with gzip.open(out, "rt") as fh:
                for ln in fh:
                    n_rows += 1
                    if ln.startswith(CHECK + "\t") and len(vals) < b - a:
                        f = ln.split("\t")
                        if a <= int(f[1]) < b:
                            vals.append(int(f[3]))
            ok = vals == want.tolist() and n_rows == bases
            n_chk = 1
        rate, n = py_rate(lambda m: O.py_wps(fetched(max(m - 2500 - 180, 0), m + 2500 + 180), max(m - 2500, 0), m + 2500, sizes[CHECK]), mids21, cap_s=5.0)
        leg.update(sites=n_site_rows, bases_scored=bases, file_MB=round(os.path.getsize(out) / 1e6, 1),
                   sites_per_s=round(n_site_rows / leg["best_s"], 1), results_ok=bool(ok and n_chk > 0),
                   checked=f"contig {CHECK}: {n_chk} site window(s) read back from the file == C oracle WPS" + ("" if suffix == ".bw" else f"; {bases} rows"),
                   reference_shaped_python_sites_per_s=round(rate, 3), python_sample=f"{n} sites of contig {CHECK}",
                   x_vs_reference_shaped_python=round(n_site_rows / leg["best_s"] / rate, 1))
        floor(leg, *out_floor[suffix])
        res[key] = leg
        os.remove(out)

    # ---------------------------------------------------------------- frag_length_intervals -> .bed
    out_iv = os.path.join(tmp, "legs_len.bed")
    leg, got = timed(lambda: frag.frag_length_intervals(genome_file, bed, out_iv, workers=threads),
                     lambda: getattr(FL, "LAST_STAGE_S", {}))
    hist, _ = O.c_fraglen_hist(fr, ws21, we21, 0, 1001, mapq_min=MAPQ, min_len=0)
    mine = [r for r in got if r[0] == CHECK]
    ok = len(got) == n_win and len(mine) == len(ws21)
    for r, h in zip(mine[::7], hist[::7]):
        w = O.py_frag_length_stats({int(k): int(h[k]) for k in np.nonzero(h)[0]}, 150)
        ok = ok and r[9] == w[5] and r[7:9] == w[3:5] and abs(r[4] - w[0]) <= 1e-9 * abs(w[0]) and r[5] == w[1] \
            and abs(r[6] - w[2]) <= 1e-9 * abs(w[2]) and abs(r[10] - w[6]) <= 1e-12
    rate, n = py_rate(lambda k: O.py_frag_length_stats(O.py_distribution(fetched(int(ws21[k]), int(we21[k])), int(ws21[k]), int(we21[k]), 0, None, "midpoint", MAPQ)),
                      range(len(ws21)))
    leg.update(windows=n_win, windows_per_s=round(n_win / leg["best_s"], 1), results_ok=bool(ok),
               checked=f"contig {CHECK}: every 7th of {len(ws21)} rows == statistics of the C oracle's histogram",
               reference_shaped_python_windows_per_s=round(rate, 2), python_sample=f"{n} windows of contig {CHECK}",
               x_vs_reference_shaped_python=round(n_win / leg["best_s"] / rate, 1))
    floor(leg)
    res["frag_length_intervals_bed"] = leg

    # ---------------------------------------------------------------- genome-wide frag_length_bins -> .tsv
    out_tsv = os.path.join(tmp, "legs_bins.tsv")
    leg, got = timed(lambda: frag.frag_length_bins(genome_file, output_file=out_tsv, summary_stats=True, short_fraction=150),
                     lambda: getattr(FL, "LAST_STAGE_S", {}))
    bins, counts = got
    nz = np.nonzero(hist_all)[0]
    ok = int(bins[0]) == int(nz[0]) and int(bins[-1]) == int(nz[-1]) and list(counts) == hist_all[nz[0]:nz[-1] + 1].tolist()
    n_rows_py = 200_000
    t0 = time.perf_counter()
    O.py_distribution(list(zip(s[:n_rows_py].tolist(), e[:n_rows_py].tolist(), q[:n_rows_py].tolist(), st[:n_rows_py].tolist())), None, None, 0, None, "midpoint", MAPQ)
    py_frag_s = n_rows_py / (time.perf_counter() - t0)
    n_frag = sum(synth.n_fragments(sizes[c], depth) for c in names)
    leg.update(fragments=n_frag, fragments_per_s_M=round(n_frag / leg["best_s"] / 1e6, 1), results_ok=bool(ok),
               checked=f"all {len(counts)} bins == the length distribution of the generated mapq >= {MAPQ} fragments ({int(hist_all.sum())})",
               reference_shaped_python_fragments_per_s_M=round(py_frag_s / 1e6, 3), python_sample=f"{n_rows_py} rows of contig {CHECK}",
               x_vs_reference_shaped_python=round(n_frag / leg["best_s"] / py_frag_s, 1))
    floor(leg)
    res["frag_length_bins_genome"] = leg
    source.close_all()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="contig lengths x scale (1.0 = b37)")
    ap.add_argument("--sites", type=int, default=20_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    import shutil
    import tempfile
    import torch
    from finaletoolkit_amd import _lib, source, writers
    _lib.load()
    dev = torch.device("cuda", 0)
    sizes = {c: max(int(n * args.scale), 200_000) for c, n in synth.B37_SIZES.items()}
    tmp = tempfile.mkdtemp(prefix="ftk_legs_")
    try:
        path = os.path.join(tmp, "genome.frag.gz")
        t0 = time.perf_counter()
        names = list(sizes)
        text_bytes = 0
        for k, c in enumerate(names):
            s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, sizes[c], synth.n_fragments(sizes[c], 30.0), synth.SEED_BASE + k))
            with writers.frag_rows(c, s, e, q, st) as text:
                writers.bgzf_write(path, text, 1, append=k > 0, write_eof=k == len(names) - 1)
                text_bytes += text.n
        open(path + ".tbi", "wb").close()
        sys.stderr.write(f"genome file {os.path.getsize(path) / 1e9:.2f} GB written in {time.perf_counter() - t0:.1f} s\n")
        import bench  # the decode floor of bench.py's file legs: compressed bytes over PCIe, the inflate kernel alone
        h2d, rates = bench.measure_h2d_gbs(torch, dev), bench.inflate_alone_rates()
        probe = dict(best_s=1.0, median_s=1.0)
        bench.leg_floor(probe, os.path.getsize(path), text_bytes, "text", h2d, rates)
        res = measure(torch, tmp, path, sizes, source.usable_cores(), raw_floor=probe.get("floor"), reps=args.reps, n_sites=args.sites)
        text = json.dumps(res, indent=1)
        print(text)
        if args.out:
            open(args.out, "w").write(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
