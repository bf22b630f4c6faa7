"""
ctypes binding of ``libftk_hip.so`` (C ABI: ``include/ftk.h``).

The shared object is built in-tree by ``finaletoolkit_amd/csrc/Makefile``
(``__graft_entry__.build()`` drives it).  There is no fallback: if the library
is missing, or no MI355X is usable, the callers raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FTK_LIB") or os.path.join(_HERE, "libftk_hip.so")  # FTK_LIB: experiments with alternative builds
CSRC = os.path.join(_HERE, "csrc")

FTK_OK = 0
FTK_ERR_INVALID = -1
FTK_ERR_NO_DEVICE = -2
FTK_ERR_HIP = -3
FTK_ERR_OOM = -4
FTK_ERR_IO = -5
FTK_ERR_FORMAT = -6
FTK_ERR_NO_CONTIG = -7
FTK_ERR_UNSORTED = -8

OPEN_LO = -(2 ** 31)
OPEN_HI = 2 ** 31 - 1
LEN_OPEN = -1
POLICY = {"midpoint": 0, "any": 1}
POLICY_FETCH = 2  # no intersect test: the index query alone (AlignmentWrapper.fetch)
FETCH_TABIX = 0
FETCH_BAM_READ1 = 1
MAX_TELOMERES = 8

# every symbol include/ftk.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "ftk_version", "ftk_device_count", "ftk_ctx_create", "ftk_ctx_destroy", "ftk_last_error",
    "ftk_ctx_set_stream", "ftk_ctx_sync", "ftk_timer_start", "ftk_timer_stop", "ftk_event_record",
    "ftk_event_elapsed_ms",
    "ftk_frags_from_host", "ftk_frags_from_device", "ftk_frags_set_read1", "ftk_frags_set_order", "ftk_frags_load_fraggz", "ftk_frags_load_bam", "ftk_frags_name",
    "ftk_frags_info", "ftk_frags_release",
    "ftk_fragfile_decode", "ftk_bam_decode", "ftk_host_alloc", "ftk_host_alloc_pageable", "ftk_host_free", "ftk_cache_trim", "ftk_wps_async", "ftk_result_wait", "ftk_fragfile_index_contigs", "ftk_fragstream_open", "ftk_fragstream_open_device", "ftk_fragstream_open_region", "ftk_fragtable_is_device", "ftk_fragtable_ready_event", "ftk_fragtable_columns_to_host", "ftk_fragtable_read1_to_host", "ftk_fragstream_next", "ftk_fragstream_n_refs",
    "ftk_fragstream_ref_name", "ftk_fragstream_ref_length", "ftk_fragstream_close", "ftk_fragstream_stage_ms", "ftk_fragstream_skipped", "ftk_fragtable_skipped", "ftk_fragtable_error", "ftk_fragtable_is_bed6",
    "ftk_fragtable_n_contigs", "ftk_fragtable_contig_name", "ftk_fragtable_contig_length",
    "ftk_fragtable_contig_rows", "ftk_fragtable_columns", "ftk_fragtable_order", "ftk_fragtable_is_pinned", "ftk_fragtable_free",
    "ftk_frags_from_table",
    "ftk_window_counts", "ftk_delfi_counts", "ftk_fraglen_hist", "ftk_fraglen_stats", "ftk_window_features",
    "ftk_window_features_batch", "ftk_wps_batch", "ftk_wps_window_features", "ftk_window_features_wps", "ftk_frag_lengths",
    "ftk_frag_select",
    "ftk_wps", "ftk_wps_intervals", "ftk_cleavage", "ftk_cleavage_intervals", "ftk_wps_adjust",
    "ftk_ref_upload", "ftk_ref_upload_file", "ftk_ref_release", "ftk_ref_gc_counts", "ftk_ref_set_layout", "ftk_motif_counts",
    "ftk_format_wig_i64", "ftk_format_bedgraph_i64", "ftk_format_bedgraph_f64", "ftk_buffer_free", "ftk_file_write", "ftk_gzip_members",
    "ftk_bigwig_fixedstep_sections", "ftk_format_frag_rows", "ftk_bgzf_write", "ftk_synth_bam_contig", "ftk_fill_wps_records", "ftk_bgzf_inflate_device",
    "ftk_comm_unique_id", "ftk_comm_create", "ftk_comm_size", "ftk_allgather_i64", "ftk_allreduce_sum_i64", "ftk_comm_send",
    "ftk_comm_recv", "ftk_comm_join", "ftk_comm_destroy",
]


class FtkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ftk error {code}: {message}")
        self.code = code
        self.message = message


class Filter(C.Structure):
    _fields_ = [("mapq_min", C.c_int32), ("min_len", C.c_int32), ("max_len", C.c_int32),
                ("policy", C.c_int32), ("fetch_mode", C.c_int32)]


class Gaps(C.Structure):
    _fields_ = [("has_gaps", C.c_int32), ("cen_start", C.c_int32), ("cen_stop", C.c_int32),
                ("n_telo", C.c_int32), ("telo_start", C.c_int32 * MAX_TELOMERES),
                ("telo_stop", C.c_int32 * MAX_TELOMERES)]


class FeatureItem(C.Structure):
    _fields_ = [("contig_id", C.c_int32), ("n_win", C.c_int64), ("w_start", C.c_void_p), ("w_end", C.c_void_p),
                ("bl_start", C.c_void_p), ("bl_end", C.c_void_p), ("n_bl", C.c_int64), ("gaps", C.POINTER(Gaps))]


class Motif(C.Structure):
    _fields_ = [("k", C.c_int32), ("fwd_offset", C.c_int32), ("rev_offset", C.c_int32),
                ("both_strands", C.c_int32), ("negative_strand", C.c_int32), ("guard", C.c_int32),
                ("rev_oob_is_error", C.c_int32)]


_lib = None


def build(force: bool = False) -> str:
    """Compile libftk_hip.so for gfx950 with hipcc (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))]
    srcs.append(os.path.join(_HERE, "..", "include", "ftk.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` / ``libhsa-runtime64.so``
    (soname ``libamdhip64.so.7``, the one ``libftk_hip.so`` asks for).  When torch is imported first, the dynamic
    linker hands that copy to ``libftk_hip.so`` too; when ``libftk_hip.so`` comes first it would bind ``/opt/rocm``'s
    copy and a later ``import torch`` would bring a second runtime into the process, which then finds no device.
    So, if torch is installed but not imported yet, its bundled runtime is loaded first (no torch import: only
    the two shared objects).  ``FTK_SYSTEM_HIP=1`` keeps the system runtime (processes that never import torch)."""
    import sys
    if "torch" in sys.modules or os.environ.get("FTK_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def hip_runtimes_mapped():
    """Paths of every ``libamdhip64`` mapped into this process (``/proc/self/maps``)."""
    found = []
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64" in os.path.basename(path) and os.path.realpath(path) not in found:
                    found.append(os.path.realpath(path))
    except OSError:
        pass
    return found


def _check_single_hip_runtime():
    """``_share_torch_hip_runtime`` relies on torch's bundled runtime carrying the soname ``libftk_hip.so`` was linked
    against.  If it does not (a torch wheel of another ROCm major), the linker maps /opt/rocm's copy as well and the
    process holds two HIP runtimes - the "finds no device" failure.  Say so instead of failing later."""
    found = hip_runtimes_mapped()
    if len(found) > 1:
        import warnings
        warnings.warn(
            "finaletoolkit_amd: two HIP runtimes are mapped into this process (" + ", ".join(found) + "); "
            "libftk_hip.so and torch would each see their own and one of them finds no device.  Set FTK_SYSTEM_HIP=1 "
            "in processes that do not import torch, or rebuild libftk_hip.so against the ROCm version torch bundles.",
            RuntimeWarning, stacklevel=3)
    return found


def _hardware_queues():
    """See ``finaletoolkit_amd/__init__.py``: the decoder's hardware-queue count is set when the package is imported
    (before any HIP call of the process); repeated here for hosts that reach the library without the package import
    having run first in this process' environment."""
    from . import _hardware_queues as set_queues
    set_queues()


def load() -> C.CDLL:
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        try:  # fresh checkout: compile in-tree once (about a minute); never a substitute implementation
            build()
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "finaletoolkit_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    _hardware_queues()
    lib = C.CDLL(LIB_PATH)
    _check_single_hip_runtime()
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.ftk_version.restype = C.c_char_p
    lib.ftk_last_error.restype = C.c_char_p
    lib.ftk_last_error.argtypes = [vp]
    lib.ftk_fragtable_error.restype = C.c_char_p
    lib.ftk_device_count.argtypes = [C.POINTER(C.c_int)]
    pp, pi64 = C.POINTER(C.c_void_p), C.POINTER(C.c_int64)
    lib.ftk_format_wig_i64.argtypes = [vp, i64, C.c_int, pp, pi64]
    lib.ftk_format_bedgraph_i64.argtypes = [C.c_char_p, vp, vp, i64, vp, C.c_int, pp, pi64]
    lib.ftk_format_bedgraph_f64.argtypes = [C.c_char_p, vp, vp, i64, vp, C.c_int, pp, pi64]
    lib.ftk_buffer_free.argtypes = [vp]
    lib.ftk_buffer_free.restype = None
    lib.ftk_format_frag_rows.argtypes = [C.c_char_p, vp, vp, vp, vp, i64, C.c_int, C.c_int, pp, pi64]
    lib.ftk_bgzf_inflate_device.argtypes = [vp, vp, i64, vp, i64, pi64]
    lib.ftk_fill_wps_records.argtypes = [vp, i64, vp, i64, vp, C.c_int]
    lib.ftk_bgzf_write.argtypes = [C.c_char_p, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.ftk_synth_bam_contig.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, i64, i32, i32, C.c_uint64, C.c_int, C.c_int,
                                         vp, i64, pi64, pi64, pi64]
    lib.ftk_file_write.argtypes = [C.c_char_p, vp, i64, C.c_int, C.c_int, C.c_int]
    lib.ftk_gzip_members.argtypes = [vp, i64, C.c_int, C.c_int, pp, pi64]
    lib.ftk_bigwig_fixedstep_sections.argtypes = [C.c_uint32, vp, vp, i64, vp, C.c_int, i32, C.c_int, C.c_int, pp, pi64,
                                                  pi64, pp, pp]
    lib.ftk_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.ftk_ctx_destroy.argtypes = [vp]
    lib.ftk_ctx_destroy.restype = None
    lib.ftk_ctx_set_stream.argtypes = [vp, vp]
    lib.ftk_ctx_sync.argtypes = [vp]
    lib.ftk_timer_start.argtypes = [vp]
    lib.ftk_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ftk_event_record.argtypes = [vp, C.c_int]
    lib.ftk_event_elapsed_ms.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.ftk_frags_from_host.argtypes = [vp, C.c_int, vp, vp, vp, vp, i64]
    lib.ftk_frags_from_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, i64]
    lib.ftk_frags_set_read1.argtypes = [vp, C.c_int, vp, vp, i64]
    lib.ftk_frags_set_order.argtypes = [vp, C.c_int, vp, i64]
    lib.ftk_fragtable_order.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.ftk_frags_load_fraggz.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.ftk_frags_load_bam.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.ftk_frags_name.argtypes = [vp, C.c_int]
    lib.ftk_frags_name.restype = C.c_char_p
    lib.ftk_frags_info.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    lib.ftk_frags_release.argtypes = [vp, C.c_int]
    lib.ftk_fragfile_decode.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.ftk_bam_decode.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.ftk_host_alloc.argtypes = [C.c_int64, C.POINTER(vp)]
    lib.ftk_host_alloc_pageable.argtypes = [C.c_int64, C.POINTER(vp)]
    lib.ftk_host_free.argtypes = [vp]
    lib.ftk_host_free.restype = None
    lib.ftk_cache_trim.argtypes = []
    lib.ftk_cache_trim.restype = C.c_int64
    lib.ftk_fragfile_index_contigs.argtypes = [C.c_char_p, C.c_char_p, i64, C.POINTER(i64), C.POINTER(C.c_int)]
    lib.ftk_fragstream_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.ftk_fragstream_open_device.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.ftk_fragstream_open_region.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(vp)]
    lib.ftk_fragtable_is_device.argtypes = [vp, C.c_int]
    lib.ftk_fragtable_ready_event.argtypes = [vp, C.c_int]
    lib.ftk_fragtable_ready_event.restype = vp
    lib.ftk_fragtable_columns_to_host.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.ftk_fragtable_read1_to_host.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.ftk_fragstream_next.argtypes = [vp, C.POINTER(vp)]
    lib.ftk_fragstream_n_refs.argtypes = [vp]
    lib.ftk_fragstream_ref_name.argtypes = [vp, C.c_int]
    lib.ftk_fragstream_ref_name.restype = C.c_char_p
    lib.ftk_fragstream_ref_length.argtypes = [vp, C.c_int]
    lib.ftk_fragstream_ref_length.restype = i64
    lib.ftk_fragstream_close.argtypes = [vp]
    lib.ftk_fragstream_stage_ms.argtypes = [vp, C.POINTER(C.c_double * 6)]
    lib.ftk_fragstream_skipped.argtypes = [vp, C.POINTER(i64 * 2)]
    lib.ftk_fragtable_skipped.argtypes = [vp, C.POINTER(i64 * 2)]
    lib.ftk_fragstream_close.restype = None
    lib.ftk_fragtable_is_bed6.argtypes = [vp]
    lib.ftk_fragtable_n_contigs.argtypes = [vp]
    lib.ftk_fragtable_contig_name.argtypes = [vp, C.c_int]
    lib.ftk_fragtable_contig_name.restype = C.c_char_p
    lib.ftk_fragtable_contig_length.argtypes = [vp, C.c_int]
    lib.ftk_fragtable_contig_length.restype = i64
    lib.ftk_fragtable_contig_rows.argtypes = [vp, C.c_int]
    lib.ftk_fragtable_contig_rows.restype = i64
    lib.ftk_fragtable_columns.argtypes = [vp, C.c_int] + [C.POINTER(vp)] * 6
    lib.ftk_fragtable_is_pinned.argtypes = [vp, C.c_int]
    lib.ftk_frags_from_table.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.ftk_fragtable_free.argtypes = [vp]
    lib.ftk_fragtable_free.restype = None
    lib.ftk_window_counts.argtypes = [vp, C.c_int, vp, vp, i64, C.POINTER(Filter), vp]
    lib.ftk_delfi_counts.argtypes = [vp, C.c_int, vp, vp, i64, i32, vp, vp, i64, C.POINTER(Gaps), vp, vp, vp]
    lib.ftk_fraglen_hist.argtypes = [vp, C.c_int, vp, vp, i64, C.POINTER(Filter), i32, i32, vp, vp]
    lib.ftk_fraglen_stats.argtypes = [vp, C.c_int, vp, vp, i64, C.POINTER(Filter), i32, i32, i32, vp]
    lib.ftk_window_features.argtypes = [vp, C.c_int, vp, vp, i64, C.POINTER(Filter), vp, i32, i32, vp, vp, i32, vp, vp,
                                        i64, C.POINTER(Gaps), vp, vp]
    lib.ftk_window_features_wps.argtypes = [vp, C.c_int, vp, vp, i64, C.POINTER(Filter), vp, i32, i32, vp, vp, i32, vp, vp,
                                            i64, C.POINTER(Gaps), vp, vp, i64, i64, i64, i32, i32, i32, i32, vp]
    lib.ftk_window_features_batch.argtypes = [vp, C.POINTER(FeatureItem), i32, C.POINTER(Filter), vp, i32, i32, vp, vp, i32,
                                              vp, vp]
    lib.ftk_wps_batch.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.ftk_wps_window_features.argtypes = [vp, C.c_int, i64, i64, i64, i32, i32, i32, i32, vp, i32, i32, i32,
                                            C.POINTER(Filter), vp, i32, i32, vp, vp, i32, vp, vp, i64, C.POINTER(Gaps), vp, vp]
    lib.ftk_frag_lengths.argtypes = [vp, C.c_int, i32, i32, C.POINTER(Filter), vp, i64, C.POINTER(i64)]
    lib.ftk_frag_select.argtypes = [vp, C.c_int, i32, i32, C.POINTER(Filter), vp, vp, vp, vp, i64, C.POINTER(i64)]
    lib.ftk_wps.argtypes = [vp, C.c_int, i64, i64, i64, i32, i32, i32, i32, vp]
    lib.ftk_wps_async.argtypes = [vp, C.c_int, i64, i64, i64, i32, i32, i32, i32, vp, C.POINTER(C.c_int)]
    lib.ftk_result_wait.argtypes = [vp, C.c_int]
    lib.ftk_wps_intervals.argtypes = [vp, C.c_int, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]
    lib.ftk_cleavage.argtypes = [vp, C.c_int, i64, i64, i32, i32, i32, vp]
    lib.ftk_cleavage_intervals.argtypes = [vp, C.c_int, vp, vp, i64, vp, i32, i32, i32, vp]
    lib.ftk_wps_adjust.argtypes = [vp, vp, vp, i64, i32, C.c_int, vp, i32, vp, vp, vp]
    lib.ftk_ref_upload.argtypes = [vp, C.c_int, vp, i64, C.c_int]
    lib.ftk_ref_upload_file.argtypes = [vp, C.c_int, C.c_char_p, i64, i64, C.c_int]
    lib.ftk_ref_release.argtypes = [vp, C.c_int]
    lib.ftk_ref_gc_counts.argtypes = [vp, C.c_int, vp, vp, i64, vp]
    lib.ftk_ref_set_layout.argtypes = [vp, C.c_int, i64, i32, i32, vp, vp, i64]
    lib.ftk_motif_counts.argtypes = [vp, C.c_int, C.c_int, vp, vp, i64, C.POINTER(Motif), i32, i32, vp, vp, vp]
    lib.ftk_comm_unique_id.argtypes = [C.c_char_p]
    lib.ftk_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]
    lib.ftk_comm_size.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ftk_allgather_i64.argtypes = [vp, vp, i64, vp]
    lib.ftk_allreduce_sum_i64.argtypes = [vp, vp, i64]
    lib.ftk_comm_send.argtypes = [vp, C.c_int, vp, i64]
    lib.ftk_comm_recv.argtypes = [vp, C.c_int, vp, i64]
    lib.ftk_comm_join.argtypes = [vp]
    lib.ftk_comm_destroy.argtypes = [vp]
    lib.ftk_comm_destroy.restype = None
    _lib = lib
    return lib


def ptr(a):
    """Pointer for a numpy array, an int address (device pointer) or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):  # torch tensor (device or host); caller keeps it alive
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot pass {type(a)} through the C ABI")


def make_filter(quality_threshold=30, min_length=None, max_length=None, intersect_policy="midpoint",
                fetch_mode=FETCH_TABIX) -> Filter:
    if intersect_policy not in POLICY and intersect_policy != "fetch":  # ("fetch": internal, AlignmentWrapper.fetch)
        from .exceptions import InvalidInputError
        raise InvalidInputError(f"{intersect_policy} is not a valid policy")
    mn = LEN_OPEN if min_length is None else max(int(min_length), 0)
    if max_length is None:
        mx = LEN_OPEN
    elif int(max_length) < 0:  # nothing can pass: an empty length interval
        mn, mx = 1, 0
    else:
        mx = int(max_length)
    return Filter(int(quality_threshold), mn, mx, POLICY.get(intersect_policy, POLICY_FETCH), fetch_mode)


def make_gaps(gaps) -> Gaps:
    """gaps: None or (cen_start, cen_stop, [(t0, t1), ...])  (genome/gaps.py:202-215)."""
    g = Gaps()
    if gaps is None:
        return g
    tel = list(gaps[2])
    if len(tel) > MAX_TELOMERES:
        raise ValueError(f"at most {MAX_TELOMERES} telomere intervals per contig are supported")
    g.has_gaps = 1
    g.cen_start, g.cen_stop = int(gaps[0]), int(gaps[1])
    g.n_telo = len(tel)
    for i, (a, b) in enumerate(tel):
        g.telo_start[i] = int(a)
        g.telo_stop[i] = int(b)
    return g
