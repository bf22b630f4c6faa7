import os, sys, gzip, ctypes as C, numpy as np, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from finaletoolkit_amd import synth, _lib as L
from finaletoolkit_amd.engine import Engine
d = tempfile.mkdtemp()
p = d + "/b.bam"
synth.write_paired_bam_native(p, [("x", int(sys.argv[1]))], 60.0, 31, keep=())
image = open(p, "rb").read()
text = gzip.open(p, "rb").read()
eng = Engine(0)
for lanes in ("1", "0"):
    os.environ["FTK_INFLATE_LANES"] = lanes
    out = np.zeros(len(text), np.uint8)
    n = C.c_int64()
    rc = eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(n))
    same = rc == 0 and out.tobytes() == text
    print(f"lanes={lanes}: {len(image)/1e9:.2f} GB -> {len(text)/1e9:.2f} GB rc {rc} equal {same} {eng.lib.ftk_last_error(eng.ctx)[:200]!r}", flush=True)
