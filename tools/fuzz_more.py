import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from finaletoolkit_amd import synth
from finaletoolkit_amd.engine import Engine
from oracle import oracle as O
import tests.test_gpu_parity as T
eng = Engine(0)
s, e, q, st = synth.synth_contig(T.CONTIG_LEN, depth=30.0, seed=7)
eng.load_contig("synA", s, e, q, st)
data = dict(s=s, e=e, q=q, st=st, fr=O.Frags(s, e, q, st))
bad = 0
for seed in range(10, 70):
    try:
        T.test_feature_fuzz_extreme_parameters(eng, data, seed)
        T.test_wps_and_cleavage_fuzz(eng, data, seed)
    except AssertionError as ex:
        bad += 1
        print("FAIL seed", seed, str(ex)[:300])
print("extended fuzz done, failures:", bad)

# ---- motif fuzz against the Python oracle (small contig, FASTA-text and 2bit images) ----------------
import tempfile
from tests import helpers as H
from finaletoolkit_amd.reference import ReferenceGenome
rng = np.random.default_rng(99)
L0 = 300_000
seq = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.choice(9, L0, p=[.22, .22, .22, .22, .02, .025, .025, .025, .025])]
sq = seq.tobytes().decode()
tmp = tempfile.mkdtemp()
H.write_fasta(os.path.join(tmp, "r.fa"), {"m": sq}, width=61)
H.write_2bit(os.path.join(tmp, "r.2bit"), {"m": sq})
n = 20_000
fs = np.sort(rng.integers(0, L0 - 300, n)).astype(np.int32)
fe = np.minimum(fs + rng.integers(1, 400, n), L0).astype(np.int32)
fs[:4] = [0, 0, 1, 2]; fe[:4] = [1, 3, 2, 7]
mq = rng.integers(0, 61, n).astype(np.uint8); sd = rng.integers(0, 2, n).astype(np.uint8)
eng.load_contig("m", fs, fe, mq, sd)
rows = list(zip(fs.tolist(), fe.tolist(), mq.tolist(), sd.tolist()))
bad = 0
for it in range(40):
    k = int(rng.integers(1, 8)); kind = str(rng.choice(["end", "breakpoint"]))
    both, neg = [(True, False), (False, False), (False, True)][int(rng.integers(0, 3))]
    q = int(rng.choice([0, 20, 60]))
    nw = int(rng.integers(1, 12))
    ws = rng.integers(-1000, L0, nw); we = ws + rng.integers(-10, 120_000, nw)
    h = k // 2
    if kind == "breakpoint" and k % 2:
        continue
    spec = dict(fwd_offset=0, rev_offset=-k, guard=0, rev_oob_is_error=False) if kind == "end" else dict(fwd_offset=-h, rev_offset=-h, guard=h, rev_oob_is_error=False)
    want = []
    for a, b in zip(ws, we):
        try:
            want.append(O.py_region_motifs(rows, sq, int(a), int(b), k, kind, both, neg, q))
        except RuntimeError:   # end motifs, both strands, 3' k-mer off the contig: count strand-wise instead
            f1 = O.py_region_motifs([(x, y, m_, 1) for x, y, m_, _ in rows], sq, int(a), int(b), k, "end", False, False, q)
            r1 = O.py_region_motifs([r for r in rows if r[1] - k >= 0 and r[0] + k <= L0], sq, int(a), int(b), k, "end", False, True, q)
            want.append(None)
    for path in ("r.fa", "r.2bit"):
        with ReferenceGenome(os.path.join(tmp, path)) as ref:
            rid = ref.device_image(eng, "m")
            got, nf, err = eng.motif_counts("m", rid, ws, we, k, both_strands=both, negative_strand=neg, quality_threshold=q, **spec)
        for i, w_ in enumerate(want):
            if w_ is not None and not np.array_equal(got[i].astype(np.int64), w_):
                bad += 1
                print("MOTIF FAIL", it, k, kind, both, neg, q, path, int(ws[i]), int(we[i]))
print("motif fuzz done, failures:", bad)
