"""BASELINE config 5 at full contig size on one GPU: a chr1-sized 60x coordinate-sorted paired-end BAM (49.9 M pairs,
99.7 M records, > 4 GiB on disk) written in position windows (bounded memory), streamed through the product's path
(source.stream_source: BGZF inflate and record parse on the device, sort by start on the device) and scored:
coverage + 1001-bin length histogram + DELFI per 100 kb window and WPS of every base.  Checked against the C oracle
in read1-fetch mode on sampled windows / a sampled WPS range, and by the fragment count.  Stage times of the
decoder's producer thread are reported.
usage: tools/bam_big_run.py [contig_bp=249250621] [depth=60] > gpurun_out/bam_big.json"""
import json
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import bgzf, source, synth, writers  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else synth.B37_SIZES["1"]
depth = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
READ = 50
tmp = tempfile.mkdtemp(prefix="ftk_bigbam_", dir=os.environ.get("FTK_BIG_TMP"))
path = os.path.join(tmp, "big.bam")
contig = "big"

t0 = time.time()
exp = synth.write_paired_bam_native(path, [(contig, size)], depth, 4242, read_len=READ)[contig]
s, e, q, st, r1_pos, n = exp["s"], exp["e"], exp["q"], exp["st"], exp["r1s"].astype(np.int64), exp["n"]
n_records = 2 * n
res = {"contig_bp": size, "depth": depth, "pairs": n, "records": n_records, "file_GB": round(os.path.getsize(path) / 1e9, 3),
       "larger_than_4GiB": os.path.getsize(path) > (1 << 32), "write_s": round(time.time() - t0, 1),
       "threads": source.usable_cores(), "host_share": os.environ.get("FTK_BAM_HOST_SHARE", "default"), "reps": []}

from oracle import oracle as O  # noqa: E402  (checker only)
ws, we = synth.tiling_windows(size, 100_000)
for rep in range(int(os.environ.get('FTK_BIG_REPS', '3'))):
    source.close_all()
    eng = source.get_engine()
    t0 = time.perf_counter()
    for src, c in source.stream_source(path):
        t1 = time.perf_counter()
        key = src.key(c)
        r, w = eng.all_features_wps(key, ws, we, size)  # ONE launch
        t2 = t3 = time.perf_counter()
    total = t3 - t0
    st_ms = src.decode_stage_ms
    res["reps"].append({"total_s": round(total, 4), "decode_until_resident_s": round(t1 - t0, 4), "features_s": round(t2 - t1, 4),
                        "wps_and_copy_back_s": round(t3 - t2, 4), "decoder_producer_stage_ms": st_ms,
                        "fragments_per_s_M": round(n / total / 1e6, 1), "file_GB_per_s": round(os.path.getsize(path) / 1e9 / total, 2)})
    if rep == 0:  # the checks: fragment count, sampled windows and a WPS range against the oracle (read1 fetch mode)
        ok = eng.info(key)[0] == n and len(w) == size
        fr = O.Frags(s, e, q, st, r1_pos.astype(np.int32), (r1_pos + READ).astype(np.int32))
        pick = np.unique(np.concatenate([np.arange(0, len(ws), 97), [len(ws) - 1, len(ws) - 2]]))
        cov = O.c_window_counts(fr, ws[pick], we[pick], mapq_min=30)
        h, ov = O.c_fraglen_hist(fr, ws[pick], we[pick], 0, 1001, mapq_min=30)
        sh, lg, nf = O.c_delfi_counts(fr, ws[pick], we[pick], 30)
        ok = ok and bool(np.array_equal(r["coverage"][pick], cov) and np.array_equal(r["hist"][pick], h)
                         and np.array_equal(r["overflow"][pick], ov) and np.array_equal(r["short"][pick], sh)
                         and np.array_equal(r["long"][pick], lg))
        for a0 in (0, size // 2, size - 60_000):
            ok = ok and bool(np.array_equal(w[a0:a0 + 50_000], O.c_wps(fr, a0, a0 + 50_000, size)))
        res["results_ok"] = bool(ok)
        res["checked"] = f"fragment count, {len(pick)} sampled 100 kb windows (coverage, histogram, DELFI) and 3 x 50 kb of WPS against the C oracle"
    del w, r
source.close_all()
os.remove(path)
print(json.dumps(res))
