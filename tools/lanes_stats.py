"""What the super-windows of the lane-parallel inflate loop look like on a file image (library built with
-DFTK_LANES_STATS, FTK_LIB pointing at it): lanes that counted, tokens, rounds of the start-resolution, how often a window
of the older kind had to step in.  usage: FTK_LIB=.../libftk_lstats.so FTK_INFLATE_LANES=1 python3 tools/lanes_stats.py [text|bam]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import _lib as L, synth, writers  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
eng = Engine(0)
tmp = tempfile.mkdtemp(prefix="ftk_ls_")
if kind == "text":
    size = synth.B37_SIZES[os.environ.get("LS_CONTIG", "21")]
    s, e, q, st = synth.synth_contig(size, 30.0, 5)
    with writers.frag_rows("21", s, e, q, st) as rows:
        text = rows.tobytes()
    p = os.path.join(tmp, "t.gz")
    writers.bgzf_write(p, text, 1)
else:
    p = os.path.join(tmp, "b.bam")
    synth.write_paired_bam_native(p, [("x", int(os.environ.get("LS_BAM_BP", 12_000_000)))], 60.0, 31, keep=())
    import gzip
    text = gzip.open(p, "rb").read()
image = open(p, "rb").read()
out = np.zeros(len(text), np.uint8)
n = C.c_int64()
st = (C.c_ulonglong * 32)()
eng.lib.ftk_debug_lanes_stats(st, 1)
rc = eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(n))
assert rc == 0 and out.tobytes() == text
eng.lib.ftk_debug_lanes_stats(st, 0)
v = list(st)
sw = max(v[0], 1)
nb = max(v[15], 1)
print(f"per block (10 ns ticks): header + tables {v[30] / nb:.0f}, super-windows {v[31] / nb:.0f}, everything else {v[14] / nb:.0f}")
print(f"first round: trips {v[24] / max(v[0], 1):.1f}, lanes that did not fall into step {v[23] / max(v[0], 1):.2f}; lanes on the second round {v[22] / max(v[0], 1):.2f}")
print(f"{kind}: {len(text) / 1e6:.1f} MB in {-(-len(text) // 0xFF00)} blocks; super-windows {v[0]} ({len(text) / sw:.0f} bytes of output each); "
      f"lanes that counted {v[1] / sw:.1f} of 64; tokens {v[2] / sw:.0f}; trips through the decode loop: pass A {v[5] / sw:.1f}, "
      f"catch-up {v[6] / sw:.1f} in {v[3] / sw:.2f} rounds; ended at a stop {v[7] / sw:.2f}, cut at an unsettled lane {v[8] / sw:.2f}; "
      f"windows of the older kind {v[4]} ({v[4] / sw:.2f} per super-window); cycles per super-window (s_memtime, 100 MHz ticks x ?): "
      f"staging + pass A {v[10] / sw:.0f}, rounds {v[11] / sw:.0f}, compaction {v[12] / sw:.0f}, phase D {v[13] / sw:.0f}; inside D: waiting for "
      f"tokens {v[16] / sw:.0f}, places {v[17] / sw:.0f}, owners of the next group {v[18] / sw:.0f}, literals + matches (stores) {v[19] / sw:.0f}, keys {v[27] / sw:.0f}, sources + states {v[28] / sw:.0f}, doubling {v[29] / sw:.0f}, "
      f"write-behind {v[20] / sw:.0f}; groups of 64 tokens {v[25] / sw:.1f}, of them with a match that is not a plain copy {v[22] / sw:.1f}; "
      f"matches {v[21] / sw:.0f}, from HBM {v[23] / sw:.1f}, overlapping or long {v[24] / sw:.1f}; rounds of pointer doubling per group {v[26] / max(v[25], 1):.2f}")
