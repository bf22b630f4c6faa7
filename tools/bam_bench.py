#!/usr/bin/env python3
"""Host BAM decoder alone (no GPU work): a synthetic coordinate-sorted paired-end BAM -> fragment table,
whole-file and streamed, per stage (FTK_DECODE_TIMING).
usage: tools/bam_bench.py [contig=22] [depth=10] [threads=16,32]
Records are fixed-size (8-byte names, one CIGAR op, 100-base reads with random bases / qualities so the
blocks compress like sequencing data, ~45 %), built with numpy; every configuration runs in a child."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READ = 100


def write_bam(path, contig, size, depth, seed=11):
    from finaletoolkit_amd import bgzf, synth
    s, e, q, st = synth.synth_contig(size, depth, seed)
    e = np.maximum(e, s + READ)
    n = len(s)
    L = (e - s).astype(np.int64)
    fwd = st == 1
    # read1 carries the fragment's strand; its mate sits at the other end
    r1_pos = np.where(fwd, s, e - READ).astype(np.int64)
    r2_pos = np.where(fwd, e - READ, s).astype(np.int64)
    rec = np.dtype([("block_size", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"),
                    ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_ref", "<i4"),
                    ("next_pos", "<i4"), ("tlen", "<i4"), ("name", "S8"), ("cigar", "<u4"),
                    ("seq", "u1", (READ // 2,)), ("qual", "u1", (READ,))])
    a = np.zeros(2 * n, rec)
    a["block_size"] = rec.itemsize - 4
    a["l_name"] = 8
    a["n_cigar"] = 1
    a["l_seq"] = READ
    a["cigar"] = READ << 4
    a["pos"][:n], a["pos"][n:] = r1_pos, r2_pos
    a["next_pos"][:n], a["next_pos"][n:] = r2_pos, r1_pos
    a["mapq"][:n] = a["mapq"][n:] = q
    a["tlen"][:n] = np.where(fwd, L, -L)
    a["tlen"][n:] = np.where(fwd, -L, L)
    a["flag"][:n] = np.where(fwd, 99, 83)
    a["flag"][n:] = np.where(fwd, 147, 163)
    ids = np.char.zfill(np.arange(n).astype("U7"), 7).astype("S8")
    a["name"][:n] = a["name"][n:] = ids
    rng = np.random.default_rng(seed)
    a["seq"] = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81,
                                    0x82, 0x84, 0x88], np.uint8), size=(2 * n, READ // 2))
    a["qual"] = rng.choice(np.array([2, 11, 25, 37], np.uint8), p=[0.03, 0.07, 0.2, 0.7], size=(2 * n, READ))
    a = a[np.argsort(a["pos"], kind="stable")]
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (contig.encode(), size)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    head += struct.pack("<i", len(contig) + 1) + contig.encode() + b"\0" + struct.pack("<i", size)
    data = head + a.tobytes()
    bgzf.write_bgzf(path, data, level=1)
    return n, len(data), np.sort(s), int(e.sum())


def child(path, threads, want_rows):
    from finaletoolkit_amd import _lib as L
    lib = L.load()
    out = {"whole": [], "stream": []}
    for _ in range(3):
        t0 = time.perf_counter()
        table = C.c_void_p()
        assert lib.ftk_bam_decode(path.encode(), None, threads, C.byref(table)) == 0
        rows = sum(lib.ftk_fragtable_contig_rows(table, i) for i in range(lib.ftk_fragtable_n_contigs(table)))
        out["whole"].append(round(time.perf_counter() - t0, 4))
        lib.ftk_fragtable_free(table)
        assert rows == want_rows, (rows, want_rows)
        t0 = time.perf_counter()
        s = C.c_void_p()
        assert lib.ftk_fragstream_open(path.encode(), None, 1, threads, 2, C.byref(s)) == 0
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
        assert got == want_rows
    print(json.dumps(out))


def main():
    from finaletoolkit_amd import _lib as L, synth
    contig = sys.argv[1] if len(sys.argv) > 1 else "22"
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    thread_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,32").split(",")]
    path = os.path.join(tempfile.mkdtemp(), "synth.bam")
    t0 = time.time()
    n, raw_bytes, starts, end_sum = write_bam(path, contig, synth.B37_SIZES[contig], depth)
    print(json.dumps({"pairs": n, "records": 2 * n, "bam_MB_uncompressed": round(raw_bytes / 1e6, 1),
                      "file_MB": round(os.path.getsize(path) / 1e6, 1), "write_s": round(time.time() - t0, 1)}), flush=True)
    # the decoded columns are the fragments the records were built from
    lib = L.load()
    table = C.c_void_p()
    assert lib.ftk_bam_decode(path.encode(), None, 4, C.byref(table)) == 0
    ps = [C.c_void_p() for _ in range(6)]
    assert lib.ftk_fragtable_columns(table, 0, *[C.byref(p) for p in ps]) == 0
    got_s = np.ctypeslib.as_array(C.cast(ps[0], C.POINTER(C.c_int32)), (n,))
    got_e = np.ctypeslib.as_array(C.cast(ps[1], C.POINTER(C.c_int32)), (n,))
    assert lib.ftk_fragtable_contig_rows(table, 0) == n and np.array_equal(got_s, starts)
    assert int(got_e.astype(np.int64).sum()) == end_sum
    lib.ftk_fragtable_free(table)
    for nodeflate in ("", "1"):
        for th in thread_list:
            env = dict(os.environ, FTK_DECODE_TIMING="1")
            if nodeflate:
                env["FTK_NO_LIBDEFLATE"] = "1"
            r = subprocess.run([sys.executable, __file__, "--child", path, str(th), str(n)], env=env,
                               capture_output=True, text=True)
            stages = [ln for ln in r.stderr.splitlines() if "[ftk" in ln][-12:]
            print(("zlib" if nodeflate else "libdeflate"), "threads", th, r.stdout.strip() or r.stderr[-400:], flush=True)
            for ln in stages:
                print("    ", ln)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        main()
