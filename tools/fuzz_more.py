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
