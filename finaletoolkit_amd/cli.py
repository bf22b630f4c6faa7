"""
Command line for the hot-path commands, flag-compatible with the reference's
``finaletoolkit`` CLI (``cli/commands/__init__.py:92-127,130-232,280-324,415-479`` and ``cli/_args.py``):
``coverage``, ``frag-length-bins``, ``frag-length-intervals``, ``wps``, ``delfi`` (+ ``cleavage-profile``,
``adjust-wps``, ``agg-bw``, ``gap-bed``, ``end-motifs``, ``interval-end-motifs``, ``breakpoint-motifs``, ``interval-breakpoint-motifs``,
``mds``, ``regional-mds``).

    python -m finaletoolkit_amd.cli coverage INPUT INTERVALS -o out.bed
"""
from __future__ import annotations

import argparse
import sys


def _shared(p, min_default, max_default=None, threads=True, policy=True, output_default="-"):
    """Options shared by the commands (cli/_args.py:46-160 of the reference)."""
    p.add_argument("-r", "--reference", dest="reference_file", default=None, metavar="FASTA_OR_2BIT")
    p.add_argument("-o", "--output", dest="output_file", default=output_default, metavar="FILE")
    p.add_argument("--min-length", dest="min_length", type=int, default=min_default, metavar="BP")
    p.add_argument("--max-length", dest="max_length", type=int, default=max_default, metavar="BP")
    if policy:
        p.add_argument("-p", "--intersect-policy", dest="intersect_policy", choices=["midpoint", "any"],
                       default="midpoint")
    p.add_argument("-q", "--min-mapq", dest="quality_threshold", type=int, default=30, metavar="N")
    if threads:
        p.add_argument("-t", "--threads", dest="workers", type=int, default=1, metavar="N")
    p.add_argument("-v", "--verbose", action="count", default=0)


# commands whose work is dealt to the ranks of a process group (the reference's Pool(workers) fan-outs:
# frag/_coverage.py:212-248, _delfi.py:289-300, _multi_wps.py:196-198, _frag_length.py:571-593,
# _cleavage_profile.py:372, _motif_common.py:635-685); every other command is one process on one GPU
SHARDED_COMMANDS = ("coverage", "delfi", "wps", "cleavage-profile", "frag-length-bins", "frag-length-intervals",
                    "end-motifs", "interval-end-motifs", "breakpoint-motifs", "interval-breakpoint-motifs")


class _Parser(argparse.ArgumentParser):
    """argparse with the visible manners of the reference's Click group (its tests/test_cli.py:196-224): ``Usage:``
    capitalised, an unknown subcommand answered with ``No such command``, exit status 2 for both."""

    def format_usage(self):
        return super().format_usage().replace("usage:", "Usage:", 1)

    def format_help(self):
        return super().format_help().replace("usage:", "Usage:", 1)

    def error(self, message):
        import re
        m = re.match(r"argument command: invalid choice: '([^']*)'", message)
        if m:
            message = f"No such command '{m.group(1)}'."
        self.print_usage(sys.stderr)
        self.exit(2, f"Error: {message}\n")


def build_parser() -> argparse.ArgumentParser:
    from . import __version__
    ap = _Parser(prog="finaletoolkit-amd", description="MI355X fragment-feature engine")
    ap.add_argument("--version", action="version",
                    version=f"FinaleToolkit-AMD, version {__version__} (the MI355X hot path behind FinaleToolkit's interface)")
    ap.add_argument("--gpus", type=int, default=1, metavar="N",
                    help="run the command as N ranks, one per MI355X: contigs are dealt to the ranks, every rank "
                         "decodes and computes its own, the per-bin / per-interval vectors meet in one RCCL "
                         "all-gather (per-base outputs: compressed sections sent to rank 0) and rank 0 writes.  "
                         "Commands: " + ", ".join(SHARDED_COMMANDS) + "; any other command refuses N > 1.  Also "
                         "honoured under torchrun")
    sub = ap.add_subparsers(dest="command", required=True)

    p = sub.add_parser("coverage", help="fragment coverage over BED intervals")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("interval_file", metavar="REGIONS")
    p.add_argument("-n", "--normalize", action="store_true")
    p.add_argument("--scale-factor", dest="scale_factor", type=float, default=1.0, metavar="X")
    _shared(p, min_default=0)

    p = sub.add_parser("frag-length-bins", help="binned fragment-length distribution")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("-c", "--contig", default=None)
    p.add_argument("-S", "--start", type=int, default=None)
    p.add_argument("-E", "--stop", type=int, default=None)
    p.add_argument("--bin-size", dest="bin_size", type=int, default=1)
    p.add_argument("--summary-stats", dest="summary_stats", action="store_true")
    p.add_argument("--short-threshold", dest="short_fraction", type=int, default=None)
    p.add_argument("--histogram", dest="histogram_path", default=None)
    _shared(p, min_default=0, threads=False)

    p = sub.add_parser("frag-length-intervals", help="fragment-length statistics per BED interval")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("interval_file", metavar="REGIONS")
    p.add_argument("--short-threshold", dest="short_reads", type=int, default=150)
    _shared(p, min_default=0)

    p = sub.add_parser("wps", help="windowed protection score over BED sites")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("site_bed", metavar="REGIONS")
    p.add_argument("--chrom-sizes", dest="chrom_sizes", default=None)
    p.add_argument("-i", "--interval-size", dest="interval_size", type=int, default=5000)
    p.add_argument("-W", "--window-size", dest="window_size", type=int, default=120)
    _shared(p, min_default=120, max_default=180, policy=False)

    p = sub.add_parser("adjust-wps", help="median/mean + Savitzky-Golay adjustment of a raw WPS bigWig")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("interval_file", metavar="REGIONS")
    p.add_argument("chrom_sizes", metavar="CHROM_SIZES")
    p.add_argument("-o", "--output", dest="output_file", required=True, metavar="FILE")
    p.add_argument("-i", "--interval-size", dest="interval_size", type=int, default=5000)
    p.add_argument("-m", "--median-window-size", dest="median_window_size", type=int, default=1000)
    p.add_argument("--savgol-window-size", dest="savgol_window_size", type=int, default=21)
    p.add_argument("--savgol-poly-deg", dest="savgol_poly_deg", type=int, default=2)
    p.add_argument("--savgol", dest="savgol", action="store_true", default=True)
    p.add_argument("--no-savgol", dest="savgol", action="store_false")
    p.add_argument("--mean", dest="mean", action="store_true")
    p.add_argument("--subtract-edges", dest="subtract_edges", action="store_true")
    p.add_argument("--edge-size", dest="edge_size", type=int, default=500)
    p.add_argument("-t", "--threads", dest="workers", type=int, default=1, metavar="N")
    p.add_argument("-v", "--verbose", action="count", default=0)

    p = sub.add_parser("agg-bw", help="aggregate a bigWig signal over constant-length, strand-annotated BED intervals")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("interval_file", metavar="REGIONS")
    p.add_argument("-o", "--output", dest="output_file", required=True, metavar="FILE")
    p.add_argument("-m", "--median-window-size", dest="median_window_size", type=int, default=1, metavar="BP")
    p.add_argument("--mean", dest="mean", action="store_true")
    p.add_argument("-v", "--verbose", action="count", default=0)

    p = sub.add_parser("gap-bed", help="BED4 of centromeres, telomeres and short-arm intervals of a reference genome")
    p.add_argument("reference_genome", metavar="GENOME", choices=["hg19", "b37", "human_g1k_v37", "hg38", "GRCh38"])
    p.add_argument("output_file", metavar="OUTPUT")

    p = sub.add_parser("cleavage-profile", help="cleavage proportion over BED intervals")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("interval_file", metavar="REGIONS")
    p.add_argument("chrom_sizes", metavar="CHROM_SIZES")
    p.add_argument("--pad-left", dest="left", type=int, default=0)
    p.add_argument("--pad-right", dest="right", type=int, default=0)
    _shared(p, min_default=0, policy=False)
    p.set_defaults(quality_threshold=20)

    for name, kdef, with_bed in (("end-motifs", 4, False), ("interval-end-motifs", 4, True),
                                 ("breakpoint-motifs", 6, False), ("interval-breakpoint-motifs", 6, True)):
        p = sub.add_parser(name, help=f"k-mer {name.replace('interval-', '')} frequencies"
                                      + (" per BED interval" if with_bed else ""))
        p.add_argument("input_file", metavar="INPUT")
        p.add_argument("refseq_file", metavar="REFERENCE")
        if with_bed:
            p.add_argument("intervals", metavar="REGIONS")
        p.add_argument("-k", "--kmer-length", dest="k", type=int, default=kdef, metavar="K")
        p.add_argument("--min-length", dest="min_length", type=int, default=50, metavar="BP")
        p.add_argument("--max-length", dest="max_length", type=int, default=None, metavar="BP")
        p.add_argument("--strand", choices=["both", "forward", "reverse"], default="both")
        p.add_argument("-o", "--output", dest="output_file", default="-", metavar="FILE")
        p.add_argument("-q", "--min-mapq", dest="quality_threshold", type=int, default=20, metavar="N")
        p.add_argument("-t", "--threads", dest="workers", type=int, default=1, metavar="N")
        p.add_argument("-v", "--verbose", action="count", default=0)

    p = sub.add_parser("mds", help="motif diversity score of a k-mer frequency table")
    p.add_argument("file_path", metavar="INPUT", nargs="?", default="-")
    p.add_argument("-s", "--sep", default="\t")
    p.add_argument("--header", type=int, default=0)

    p = sub.add_parser("regional-mds", help="regional motif diversity score per interval")
    p.add_argument("file_path", metavar="INPUT", nargs="?", default="-")
    p.add_argument("file_out", metavar="OUTPUT")
    p.add_argument("-s", "--sep", default="\t")
    p.add_argument("--header", type=int, default=0)
    p.add_argument("--miller-madow", dest="miller_madow", action="store_true")

    p = sub.add_parser("delfi", help="DELFI short/long fragment features")
    p.add_argument("input_file", metavar="INPUT")
    p.add_argument("chrom_sizes", metavar="CHROM_SIZES")
    p.add_argument("reference_file", metavar="REFERENCE")
    p.add_argument("bins_file", metavar="BINS")
    p.add_argument("-b", "--blacklist", dest="blacklist_file", default=None)
    p.add_argument("-g", "--gap-file", dest="gap_file", default=None)
    p.add_argument("-o", "--output", dest="output_file", default="-")
    p.add_argument("--no-gc-correct", dest="no_gc_correct", action="store_true",
                   help="skip the LOESS GC correction.  The correction itself is delegated to the third-party "
                        "`loess` package exactly as in the reference (frag/_delfi_gc_correct.py:71-77); it is not "
                        "part of this engine, so without that package installed this flag is required")
    p.add_argument("--remove-nocov", dest="remove_nocov", action="store_true", default=True)
    p.add_argument("--no-remove-nocov", dest="remove_nocov", action="store_false")
    p.add_argument("--merge-bins", dest="merge_bins", action="store_true", default=True)
    p.add_argument("--no-merge-bins", dest="merge_bins", action="store_false")
    p.add_argument("--merge-size", dest="window_size", type=int, default=5000000)
    p.add_argument("-q", "--min-mapq", dest="quality_threshold", type=int, default=30)
    p.add_argument("-t", "--threads", dest="workers", type=int, default=1)
    p.add_argument("-v", "--verbose", action="count", default=0)
    return ap


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    a = build_parser().parse_args(argv)
    import os
    from . import sharding
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if (a.gpus > 1 or world_env > 1) and a.command not in SHARDED_COMMANDS:
        # never N duplicate runs writing one output file: a command that does not shard runs as ONE process
        sys.stderr.write(f"`{a.command}` does not shard over GPUs: run it without --gpus / outside torchrun "
                         f"(sharded commands: {', '.join(SHARDED_COMMANDS)})\n")
        return 2
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one rank per GPU (the reference's Pool(workers) fan-out): this process only launches them
        return sharding.launch_ranks([sys.executable, "-m", "finaletoolkit_amd.cli"] + argv, a.gpus,
                                     share_gpu=os.environ.get("FTK_SHARE_GPU") == "1")
    if a.command in SHARDED_COMMANDS:
        sharding.init_from_env()
    from . import frag
    if a.command == "coverage":
        frag.coverage(a.input_file, a.interval_file, a.output_file, scale_factor=a.scale_factor,
                      min_length=a.min_length, max_length=a.max_length, normalize=a.normalize,
                      intersect_policy=a.intersect_policy, quality_threshold=a.quality_threshold, workers=a.workers,
                      verbose=a.verbose, reference_file=a.reference_file)
    elif a.command == "frag-length-bins":
        frag.frag_length_bins(a.input_file, contig=a.contig, start=a.start, stop=a.stop, min_length=a.min_length,
                              max_length=a.max_length, bin_size=a.bin_size, output_file=a.output_file,
                              intersect_policy=a.intersect_policy, quality_threshold=a.quality_threshold,
                              summary_stats=a.summary_stats, short_fraction=a.short_fraction,
                              histogram_path=a.histogram_path, verbose=a.verbose, reference_file=a.reference_file)
    elif a.command == "frag-length-intervals":
        frag.frag_length_intervals(a.input_file, a.interval_file, output_file=a.output_file, min_length=a.min_length,
                                   max_length=a.max_length, quality_threshold=a.quality_threshold,
                                   intersect_policy=a.intersect_policy, short_reads=a.short_reads, workers=a.workers,
                                   verbose=a.verbose, reference_file=a.reference_file)
    elif a.command == "wps":
        frag.multi_wps(a.input_file, a.site_bed, chrom_sizes=a.chrom_sizes, output_file=a.output_file,
                       window_size=a.window_size, interval_size=a.interval_size, min_length=a.min_length,
                       max_length=a.max_length, quality_threshold=a.quality_threshold, workers=a.workers,
                       verbose=a.verbose, reference_file=a.reference_file)
    elif a.command == "adjust-wps":
        frag.adjust_wps(a.input_file, a.interval_file, a.output_file, a.chrom_sizes, interval_size=a.interval_size,
                        median_window_size=a.median_window_size, savgol_window_size=a.savgol_window_size,
                        savgol_poly_deg=a.savgol_poly_deg, savgol=a.savgol, mean=a.mean,
                        subtract_edges=a.subtract_edges, edge_size=a.edge_size, workers=a.workers,
                        verbose=a.verbose)
    elif a.command == "agg-bw":
        from .utils import agg_bw
        agg_bw(a.input_file, a.interval_file, a.output_file, median_window_size=a.median_window_size, mean=a.mean,
               verbose=a.verbose)
    elif a.command == "gap-bed":
        from .genome.gaps import _cli_gap_bed
        _cli_gap_bed(a.reference_genome, a.output_file)
    elif a.command == "cleavage-profile":
        frag.multi_cleavage_profile(a.input_file, a.interval_file, a.chrom_sizes, left=a.left, right=a.right,
                                    min_length=a.min_length, max_length=a.max_length,
                                    quality_threshold=a.quality_threshold, output_file=a.output_file,
                                    workers=a.workers, verbose=a.verbose, reference_file=a.reference_file)
    elif a.command in ("end-motifs", "interval-end-motifs", "breakpoint-motifs", "interval-breakpoint-motifs"):
        # --strand -> both_strands / negative_strand (cli/_dispatch.py:_translate_strand of the reference)
        fn = getattr(frag, a.command.replace("-", "_"))
        args = [a.input_file, a.refseq_file] + ([a.intervals] if a.command.startswith("interval") else [])
        fn(*args, k=a.k, min_length=a.min_length, max_length=a.max_length, both_strands=a.strand == "both",
           negative_strand=a.strand == "reverse", output_file=a.output_file,
           quality_threshold=a.quality_threshold, workers=a.workers, verbose=a.verbose)
    elif a.command == "mds":
        from .frag._end_motifs import _cli_mds
        _cli_mds(a.file_path, a.sep, a.header)
    elif a.command == "regional-mds":
        from .frag._end_motifs import _cli_regional_mds
        _cli_regional_mds(a.file_path, a.file_out, a.sep, a.header, a.miller_madow)
    elif a.command == "delfi":
        frag.delfi(a.input_file, a.chrom_sizes, a.bins_file, a.reference_file, blacklist_file=a.blacklist_file,
                   gap_file=a.gap_file, output_file=a.output_file, no_gc_correct=a.no_gc_correct,
                   remove_nocov=a.remove_nocov, merge_bins=a.merge_bins, window_size=a.window_size,
                   quality_threshold=a.quality_threshold, workers=a.workers, verbose=a.verbose)
    sharding.finalize()
    return 0


if __name__ == "__main__":
    sys.exit(main())
