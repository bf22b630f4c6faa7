import os, sys, time, tempfile, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from finaletoolkit_amd import synth, _lib as L
from finaletoolkit_amd.engine import Engine
d=tempfile.mkdtemp()
exp=synth.write_paired_bam(d+"/x.bam","mid",12_000_000,60.0,31)
image=open(d+"/x.bam","rb").read()
eng=Engine(0)
n=C.c_int64(); out=np.zeros(1,np.uint8)
rc=eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), 0, C.byref(n))
out=np.zeros(n.value,np.uint8)
for rep in range(3):
    t=time.perf_counter(); rc=eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(n)); dt=time.perf_counter()-t
    assert rc==0 or os.environ.get("INFLATE_BENCH_NOCHECK") == "1", eng.lib.ftk_last_error(eng.ctx)
    print(f"BAM image {len(image)/1e6:.0f} MB -> {n.value/1e6:.0f} MB: call {dt*1e3:.1f} ms", flush=True)
import zlib
# host check of the first MB
print("head ok", bytes(out[:4])==b"BAM\x01")
