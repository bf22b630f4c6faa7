import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from finaletoolkit_amd import bgzf, source, synth
names=["19","20","21","22"]
tmp=tempfile.mkdtemp(); dev=torch.device("cuda",0); rows=[]
for k,c in enumerate(names):
    size=synth.B37_SIZES[c]
    s,e,q,st=(t.cpu().numpy() for t in synth.gen_contig_device(torch,dev,size,synth.n_fragments(size,30.0),900+k))
    rows.append((c,s,e,q,st))
path=os.path.join(tmp,"g.frag.gz"); bgzf.write_frag_gz(path,rows,level=1,with_index=True)
for rep in range(3):
    source.close_all()
    t0=time.perf_counter(); src=source.open_source(path); t1=time.perf_counter()
    ts=[]
    for c in names:
        a=time.perf_counter(); src.require(c); ts.append(round((time.perf_counter()-a)*1e3,1))
    print("open %.1f ms, per contig"%((t1-t0)*1e3), ts, flush=True)
source.close_all()
t0=time.perf_counter()
for s_,c in source.stream_source(path): pass
print("stream_source whole file %.1f ms"%((time.perf_counter()-t0)*1e3))
