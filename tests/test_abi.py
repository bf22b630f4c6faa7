"""
CPU: the C-ABI library loads, exports every symbol include/ftk.h declares, its
host-only decoders agree with an independent parser, and the product refuses
to run without a GPU instead of falling back.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from finaletoolkit_amd import _lib as L
from finaletoolkit_amd import bgzf, synth
from tests.helpers import DATA, GOLDEN, ROOT, read_frag_gz


def _decode(path, bam=False, contig=None, threads=3):
    lib = L.load()
    t = C.c_void_p()
    fn = lib.ftk_bam_decode if bam else lib.ftk_fragfile_decode
    rc = fn(path.encode(), None if contig is None else contig.encode(), threads, C.byref(t))
    if rc != 0:
        raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
    out = {}
    try:
        for i in range(lib.ftk_fragtable_n_contigs(t)):
            rows = lib.ftk_fragtable_contig_rows(t, i)
            name = lib.ftk_fragtable_contig_name(t, i).decode()
            ps = [C.c_void_p() for _ in range(6)]
            assert lib.ftk_fragtable_columns(t, i, *[C.byref(p) for p in ps]) == 0
            cols = []
            for p, ct in zip(ps, (C.c_int32, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32, C.c_int32)):
                cols.append(None if not p.value or rows == 0
                            else np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (rows,)).copy())
            out[name] = (rows, cols, lib.ftk_fragtable_contig_length(t, i))
        out["__bed6__"] = lib.ftk_fragtable_is_bed6(t)
    finally:
        lib.ftk_fragtable_free(t)
    return out


def test_header_symbols_all_exported():
    header = open(os.path.join(ROOT, "include", "ftk.h")).read()
    declared = set(re.findall(r"\b(ftk_[a-z0-9_]+)\s*\(", header))
    lib = L.load()
    assert declared == set(L.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.ftk_version().startswith(b"ftk-hip")


def test_no_gpu_is_a_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from finaletoolkit_amd.engine import Engine
    with pytest.raises(L.FtkError) as ei:
        Engine(0)
    assert ei.value.code == L.FTK_ERR_NO_DEVICE and "no CPU fallback" in ei.value.message
    from finaletoolkit_amd import frag
    with pytest.raises(L.FtkError):
        frag.single_coverage(os.path.join(DATA, "12.3444.b37.frag.gz"), "12")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "finaletoolkit_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dp, f)


@pytest.mark.parametrize("name", ["12.3444.b37.frag.gz", "12.3444.b37.frag.bed.gz"])
def test_text_decoder_matches_python_gzip(name):
    path = os.path.join(DATA, name)
    got = _decode(path)
    want = read_frag_gz(path)
    assert got["__bed6__"] == (1 if "bed" in name else 0)
    for c, cols in want.items():
        rows, g, length = got[c]
        assert rows == len(cols[0]) and length == -1
        for a, b in zip(g[:4], cols):
            assert np.array_equal(a, b)


def test_text_decoder_multiblock_multithread_and_contig_filter(tmp_path):
    d = read_frag_gz(os.path.join(GOLDEN, "synth.frag.gz"))
    got = _decode(os.path.join(GOLDEN, "synth.frag.gz"), threads=4)
    for c in d:
        for a, b in zip(got[c][1][:4], d[c]):
            assert np.array_equal(a, b)
    only = _decode(os.path.join(GOLDEN, "synth.frag.gz"), contig="chrB")
    assert set(k for k in only if not k.startswith("__")) == {"chrB"}
    # a larger file: many BGZF blocks, rows straddling block and thread-segment boundaries
    s, e, q, st = synth.synth_contig(3_000_000, depth=20.0, seed=3)
    p = str(tmp_path / "big.frag.gz")
    bgzf.write_frag_gz(p, [("c1", s, e, q, st), ("c2", s[:1000], e[:1000], q[:1000], st[:1000])], level=1)
    for threads in (1, 5):
        got = _decode(p, threads=threads)
        assert got["c1"][0] == len(s) and got["c2"][0] == 1000
        for a, b in zip(got["c1"][1][:4], (s, e, q, st)):
            assert np.array_equal(a, b)


def test_text_decoder_skips_malformed_rows_and_plain_gzip(tmp_path):
    import gzip
    text = ("#comment\n12\t100\t200\t60\t+\n12\tx\t300\t60\t+\n12\t150\t260\n\n12\t300\t420\t7\t-\textra\n"
            "13\t5\t50\t1000\t+\n12\t500\t600\t-3\t+\n")
    p = str(tmp_path / "m.frag.gz")
    with gzip.open(p, "wt") as fh:  # plain gzip, not BGZF
        fh.write(text)
    got = _decode(p)
    assert got["12"][0] == 2 and got["13"][0] == 1
    assert got["12"][1][0].tolist() == [100, 300] and got["12"][1][2].tolist() == [60, 7]
    assert got["12"][1][3].tolist() == [1, 0] and got["13"][1][2].tolist() == [255]
    with pytest.raises(RuntimeError):
        q = tmp_path / "junk.gz"
        q.write_bytes(b"this is not gzip")
        _decode(str(q))
    with pytest.raises(RuntimeError):
        _decode(str(tmp_path / "absent.gz"))


def test_bam_decoder_fixture():
    got = _decode(os.path.join(DATA, "12.3444.b37.bam"), bam=True)
    rows, cols, length = got["12"]
    want = read_frag_gz(os.path.join(DATA, "12.3444.b37.frag.gz"))["12"]
    assert rows == 17 and length == 133851895
    assert np.array_equal(cols[0], want[0]) and np.array_equal(cols[1], want[1]) and np.array_equal(cols[3], want[3])
    assert np.all(cols[4] >= cols[0]) and np.all(cols[5] <= cols[1])  # read1 inside its fragment
    assert len([k for k in got if not k.startswith("__")]) == 84  # @SQ lines


def test_cache_trim_without_a_gpu_is_a_no_op():
    """ftk_cache_trim gives idle cached blocks back; with nothing cached (and no device) it returns 0 and the
    library goes on working."""
    lib = L.load()
    assert lib.ftk_cache_trim() == 0
    t = _decode(os.path.join(DATA, "12.3444.b37.frag.gz"))
    assert t["12"][0] == 17 and lib.ftk_cache_trim() >= 0


def test_one_hip_runtime_in_the_process():
    """libftk_hip.so and torch share ONE libamdhip64 whichever is loaded first (_lib._share_torch_hip_runtime);
    _lib.hip_runtimes_mapped is what load() checks and warns about."""
    import subprocess
    import sys
    code = ("import sys, warnings; sys.path.insert(0, %r)\n"
            "warnings.simplefilter('error')\n"
            "from finaletoolkit_amd import _lib\n"
            "_lib.load()\n"
            "import torch\n"
            "found = _lib.hip_runtimes_mapped()\n"
            "assert len(found) == 1, found\n"
            "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-2000:]


def test_wps_kernels_keep_their_non_temporal_stores(tmp_path):
    """ISA contract of the built code object (no GPU needed): the NT variants of the WPS kernels write their scores
    with `global_store_dwordx4 ... nt`, the plain variants do not.  As a run-time branch the two stores differed only
    in metadata and the optimiser folded them into one plain store when the kernel body became a device function
    (round 3: WPS 175 -> 197 us per launch, the feature pass behind it 37 -> 53 us) -- nothing but the disassembly shows that."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    so = tmp_path / "libftk_hip.so"
    shutil.copy(L.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], capture_output=True, check=True)
    text = ""
    for f in sorted(tmp_path.iterdir()):
        if "gfx950" in f.name:
            text += subprocess.run([objdump, "-d", str(f)], capture_output=True, text=True, check=True).stdout
    bodies = {}
    for m in re.finditer(r"^[0-9a-f]+ <(_ZN3ftk(?:17wps_stream_kernel|20feat_then_wps_kernel)[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    assert len(bodies) >= 10, sorted(bodies)
    seen_nt = seen_plain = 0
    for name, body in bodies.items():
        flags = re.search(r"kernelI((?:Lb[01]E)+)", name).group(1)
        nt_variant = flags.endswith("Lb1E")  # the last template parameter
        n_nt = len(re.findall(r"global_store_dwordx4 .*\bnt\b", body))
        if nt_variant:
            assert n_nt >= 4, (name, n_nt)
            seen_nt += 1
        else:
            assert n_nt == 0, (name, n_nt)
            seen_plain += 1
    assert seen_nt >= 5 and seen_plain >= 5
