"""CPU: the host decoders under AddressSanitizer + UBSan (fixtures, truncated and bit-flipped inputs)."""
import os
import shutil
import subprocess

import pytest

from tests.helpers import DATA, ROOT


def test_decoder_under_asan_ubsan(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "decode_sanitize")
    cmd = [hipcc, "-x", "c++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", f"-I{ROOT}/include", f"-I{ROOT}/finaletoolkit_amd/csrc", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "native", "decode_sanitize.cpp"),
           os.path.join(ROOT, "finaletoolkit_amd", "csrc", "ftk_decode.cpp"),
           "-o", exe, "-lz", "-lpthread", "-ldl", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # only a missing sanitizer runtime excuses the build; a compile / link error in our sources must fail
        if any(k in r.stderr for k in ("libclang_rt", "unsupported option '-fsanitize", "cannot find -lasan", "cannot find -ltsan")):
            pytest.skip("sanitizer runtime unavailable: " + r.stderr[-300:])
        pytest.fail("sanitizer build failed: " + r.stderr[-1500:])
    # 1 MB pieces: the harness opens several hundred streams, and each would otherwise map and fault
    # a 48 MB piece buffer in (minutes of kernel time under the sanitizer's allocator)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               FTK_STREAM_PIECE=str(1 << 20),
               FTK_BAM_STRETCH="256")  # the BAM record chain is entered speculatively every 256 bytes
    r = subprocess.run([exe, DATA, str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "decode_sanitize ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]


def test_decoder_pipeline_under_tsan(tmp_path):
    """The streaming decoder's threads (producer, read-ahead, packer, worker pool) under ThreadSanitizer on
    multi-piece text and BAM files."""
    import numpy as np
    from finaletoolkit_amd import bgzf, synth
    from tests.helpers import write_synthetic_bam
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "decode_tsan")
    cmd = [hipcc, "-x", "c++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", f"-I{ROOT}/include", f"-I{ROOT}/finaletoolkit_amd/csrc", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "native", "decode_sanitize.cpp"),
           os.path.join(ROOT, "finaletoolkit_amd", "csrc", "ftk_decode.cpp"),
           "-o", exe, "-lz", "-lpthread", "-ldl", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # only a missing sanitizer runtime excuses the build; a compile / link error in our sources must fail
        if any(k in r.stderr for k in ("libclang_rt", "unsupported option '-fsanitize", "cannot find -lasan", "cannot find -ltsan")):
            pytest.skip("sanitizer runtime unavailable: " + r.stderr[-300:])
        pytest.fail("sanitizer build failed: " + r.stderr[-1500:])
    rows = []
    for k, size in enumerate((1_500_000, 400_000, 900_000)):
        s, e, q, st = synth.synth_contig(size, depth=15.0, seed=70 + k)
        rows.append((f"c{k}", s, e, q, st))
    text = str(tmp_path / "multi.frag.gz")
    bgzf.write_frag_gz(text, rows, level=1)
    rng = np.random.default_rng(3)
    contigs = [("chrA", 1_200_000), ("chrB", 500_000)]
    frags = {}
    for name, size in contigs:
        n = size // 60
        s = np.sort(rng.integers(0, size - 700, n))
        frags[name] = (s, s + rng.integers(210, 600, n), rng.integers(0, 61, n), rng.integers(0, 2, n).astype(bool))
    bam = str(tmp_path / "multi.bam")
    write_synthetic_bam(bam, contigs, frags)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0", FTK_STREAM_PIECE="65536", FTK_BAM_STRETCH="4096")
    r = subprocess.run([exe, "--files", text, bam], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "decode_sanitize ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
