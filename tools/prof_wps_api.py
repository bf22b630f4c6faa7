import cProfile, pstats, sys, os, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from finaletoolkit_amd import bgzf, frag, source, synth
size = synth.B37_SIZES["22"]
tmp = tempfile.mkdtemp()
s, e, q, st = synth.synth_contig(size, 30.0, synth.SEED_BASE + 21)
path = os.path.join(tmp, "chr22.frag.gz")
bgzf.write_frag_gz(path, [("22", s, e, q, st)], level=1, with_index=True)
for i in range(3):
    t0 = time.perf_counter(); r = frag.wps(path, "22", 0, size, size); print("no output", time.perf_counter() - t0); 
    t0 = time.perf_counter(); del r; print("  del", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); r = frag.wps(path, "22", 0, size, size); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
del r
for i in range(2):
    t0 = time.perf_counter(); r = frag.wps(path, "22", 0, size, size, output_file=tmp + "/o.wig"); print("wig", time.perf_counter() - t0); del r
