#!/usr/bin/env python3
"""Host decoder alone (no GPU work): frag.gz -> fragment table, whole-file and streamed, per stage
(FTK_DECODE_TIMING), with libdeflate / zlib and a few thread counts.
usage: tools/decode_bench.py [contigs=19,20,21,22] [threads=8,16,32]
The file is written once; every configuration runs in a child process (the switches are read once)."""
import ctypes as C
import io
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, threads):
    from finaletoolkit_amd import _lib as L
    lib = L.load()
    out = {"whole": [], "stream": []}
    cold = os.environ.get("FTK_BENCH_COLD") == "1"

    def evict():  # clean page-cache pages of the file are dropped (the file was fsync'ed by the parent)
        if cold:
            fd = os.open(path, os.O_RDONLY)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)

    for _ in range(3):
        evict()
        t0 = time.perf_counter()
        table = C.c_void_p()
        assert lib.ftk_fragfile_decode(path.encode(), None, threads, C.byref(table)) == 0
        rows = sum(lib.ftk_fragtable_contig_rows(table, i) for i in range(lib.ftk_fragtable_n_contigs(table)))
        out["whole"].append(round(time.perf_counter() - t0, 4))
        lib.ftk_fragtable_free(table)
        evict()
        t0 = time.perf_counter()
        s = C.c_void_p()
        assert lib.ftk_fragstream_open(path.encode(), None, 0, threads, 2, C.byref(s)) == 0
        got = 0
        while True:
            t = C.c_void_p()
            assert lib.ftk_fragstream_next(s, C.byref(t)) == 0
            if not t:
                break
            got += lib.ftk_fragtable_contig_rows(t, 0)
            lib.ftk_fragtable_free(t)
        lib.ftk_fragstream_close(s)
        out["stream"].append(round(time.perf_counter() - t0, 4))
        assert got == rows
    out["rows"] = int(rows)
    print(json.dumps(out))


def main():
    import numpy as np
    import pandas as pd
    from finaletoolkit_amd import bgzf, synth
    names = (sys.argv[1] if len(sys.argv) > 1 else "19,20,21,22").split(",")
    thread_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,16,32").split(",")]
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "multi.frag.gz")
    parts = []
    for c in names:
        s, e, q, st = synth.synth_contig(synth.B37_SIZES[c], 30.0, synth.SEED_BASE + list(synth.B37_SIZES).index(c))
        buf = io.StringIO()
        pd.DataFrame({"c": c, "s": s, "e": e, "q": q, "t": np.where(st == 1, "+", "-")}).to_csv(
            buf, sep="\t", header=False, index=False)
        parts.append(buf.getvalue().encode())
    text = b"".join(parts)
    bgzf.write_bgzf(path, text, level=1)
    fd = os.open(path, os.O_RDONLY)
    os.fsync(fd)
    os.close(fd)
    print(json.dumps({"text_MB": round(len(text) / 1e6, 1), "file_MB": round(os.path.getsize(path) / 1e6, 1)}))
    del text, parts
    for nodeflate in ("", "1"):
        for th in thread_list:
            env = dict(os.environ, FTK_DECODE_TIMING="1")  # FTK_BENCH_COLD=1: evict the file before every decode
            if nodeflate:
                env["FTK_NO_LIBDEFLATE"] = "1"
            r = subprocess.run([sys.executable, __file__, "--child", path, str(th)], env=env, capture_output=True, text=True)
            stages = [ln for ln in r.stderr.splitlines() if "inflate" in ln or "parse" in ln][-3:]
            print(("zlib" if nodeflate else "libdeflate"), "threads", th, r.stdout.strip(), flush=True)
            for ln in stages:
                print("    ", ln)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        main()
