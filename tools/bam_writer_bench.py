"""How fast the synthetic paired-end BAM of BASELINE config 5 is written (synth.write_paired_bam_native: records built,
sorted and deflated by the host threads in C).  usage: python3 tools/bam_writer_bench.py [contig_bp=100000000] [depth=60]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
depth = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
d = tempfile.mkdtemp(prefix="ftk_bamw_", dir=os.environ.get("FTK_BIG_TMP"))
t = time.time()
cols = synth.synth_contig(size, depth, 1)
print(f"synth_contig (numpy): {time.time() - t:.2f} s for {len(cols[0])} fragments", flush=True)
t = time.time()
synth.write_paired_bam_native(d + "/c.bam", [("x", size)], depth, 4242, fragments=lambda k, c, n: cols, keep=())
t2 = time.time() - t
sz = os.path.getsize(d + "/c.bam")
print(f"native write: {t2:.2f} s, file {sz / 1e9:.3f} GB = {sz / 1e9 / t2:.2f} GB/s of file, "
      f"{2 * len(cols[0]) * 125 / 1e9 / t2:.2f} GB/s of records, {2 * len(cols[0]) / t2 / 1e6:.1f} M records/s")
os.remove(d + "/c.bam")
os.remove(d + "/c.bam.bai")
os.rmdir(d)
