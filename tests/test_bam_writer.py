"""CPU: the synthetic paired-end BAM writer in C (``ftk_synth_bam_contig`` behind ``synth.write_paired_bam_native``:
what BASELINE config 5's whole-genome file is written with) against the numpy writer it replaces
(``synth.write_paired_bam_contigs``) - same header, same records in the same order (everything but the random sequence
and quality bytes), a linear index that points at the same records - and against the host BAM decoder (reference
``io/alignment.py:242-268`` read1 fragments).  Pure host code: no GPU."""
import gzip
import struct

import numpy as np

from finaletoolkit_amd import synth


def _records(path, read_len=50, name_len=10):
    raw = gzip.open(path, "rb").read()  # (BGZF is multi-member gzip)
    l_text = struct.unpack("<i", raw[4:8])[0]
    off = 8 + l_text
    n_ref = struct.unpack("<i", raw[off:off + 4])[0]
    off += 4
    for _ in range(n_ref):
        ln = struct.unpack("<i", raw[off:off + 4])[0]
        off += 4 + ln + 4
    rec = 36 + name_len + 4 + (read_len + 1) // 2 + read_len
    return raw[:off], np.frombuffer(raw[off:], np.uint8).reshape(-1, rec)


def _stream_pos(path, lin):
    """Linear-index entries as positions in the INFLATED stream (virtual offset -> block start + offset in the block)."""
    data = open(path, "rb").read()
    pos, at, u = 0, {}, 0
    while pos < len(data):
        size = struct.unpack("<H", data[pos + 16:pos + 18])[0] + 1
        at[pos] = u
        u += struct.unpack("<I", data[pos + size - 4:pos + size])[0]
        pos += size
    return np.array([at[int(v) >> 16] + (int(v) & 0xFFFF) if v else -1 for v in lin])


def test_native_writer_equals_the_numpy_writer(tmp_path):
    contigs = [("a", 700_000), ("b", 2_300_000), ("c", 90_000), ("empty_ish", 200_000)]
    pn, pc = str(tmp_path / "np.bam"), str(tmp_path / "c.bam")
    e1 = synth.write_paired_bam_contigs(pn, contigs, 40.0, 4242, step=1 << 19)
    e2 = synth.write_paired_bam_native(pc, contigs, 40.0, 4242, threads=3)
    h1, r1 = _records(pn)
    h2, r2 = _records(pc)
    assert h1 == h2 and r1.shape == r2.shape
    assert np.array_equal(r1[:, :50], r2[:, :50])  # block_size .. tlen, read name, CIGAR: identical
    qual = r2[:, 75:]
    assert set(np.unique(qual)) == {2, 11, 25, 37} and 0.68 < float((qual == 37).mean()) < 0.72
    assert len(np.unique(r2[:, 50:75], axis=0)) > 0.99 * len(r2)  # sequence bytes differ from record to record
    for c, _ in contigs:
        for k in ("s", "e", "q", "st", "r1s", "r1e"):
            assert np.array_equal(e1[c][k], e2[c][k]), (c, k)
        assert e1[c]["n"] == e2[c]["n"]
        assert np.array_equal(_stream_pos(pn, e1[c]["linear"]), _stream_pos(pc, e2[c]["linear"])), c
    # a second run with another thread count: the same bytes (the generator is counter-based, not per-thread)
    pc2 = str(tmp_path / "c2.bam")
    synth.write_paired_bam_native(pc2, contigs, 40.0, 4242, threads=1)
    assert open(pc, "rb").read() == open(pc2, "rb").read()
    assert open(pc + ".bai", "rb").read() == open(pc2 + ".bai", "rb").read()


def test_native_file_through_the_host_decoder(tmp_path):
    """The host BAM decoder (whole file) hands out the writer's fragments with their read1 spans, contig by contig."""
    import ctypes as C
    from finaletoolkit_amd import _lib as L
    lib = L.load()
    contigs = [("x", 400_000), ("y", 150_000)]
    p = str(tmp_path / "w.bam")
    exp = synth.write_paired_bam_native(p, contigs, 25.0, 7)
    t = C.c_void_p()
    assert lib.ftk_bam_decode(p.encode(), None, 2, C.byref(t)) == L.FTK_OK, lib.ftk_fragtable_error()
    try:
        assert lib.ftk_fragtable_n_contigs(t) == 2
        for i, (c, size) in enumerate(contigs):
            n = lib.ftk_fragtable_contig_rows(t, i)
            assert lib.ftk_fragtable_contig_name(t, i).decode() == c and n == exp[c]["n"]
            ps = [C.c_void_p() for _ in range(6)]
            assert lib.ftk_fragtable_columns(t, i, *[C.byref(x) for x in ps]) == L.FTK_OK
            cols = [np.ctypeslib.as_array(C.cast(x, C.POINTER(ct)), (n,)) for x, ct in
                    zip(ps, (C.c_int32, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32, C.c_int32))]
            # rows come sorted by fragment start (stable in file order): compare as sorted tuples
            got = sorted(zip(*(a.tolist() for a in cols)))
            want = sorted(zip(exp[c]["s"].tolist(), exp[c]["e"].tolist(), exp[c]["q"].tolist(), exp[c]["st"].tolist(),
                              exp[c]["r1s"].tolist(), exp[c]["r1e"].tolist()))
            assert got == want, c
    finally:
        lib.ftk_fragtable_free(t)
