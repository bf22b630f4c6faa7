#!/usr/bin/env python3
"""One 256 MB device -> page-locked host copy in a torch process (its bundled HIP runtime), in different stream
states.  Findings on the MI355X box: with `AMD_LOG_LEVEL=4` every transfer of the process logs
`HSA Copy copy_engine=0x1` (one DMA engine for both directions; the system's runtime, tools/native/copy_probe.hip,
gets `rec_engine_mask 0x6`); under `rocprofv3 --kernel-trace --memory-copy-trace` the same copies show up as the
blit kernel __amd_rocclr_copyBuffer instead - in every stream state tried here."""
import ctypes as C
import os
import time

import torch

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
n = 256 << 20
dptr, d2, hptr, s1, s2, ev = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), C.c_size_t(n)) == 0
assert hip.hipMalloc(C.byref(d2), C.c_size_t(n)) == 0
assert hip.hipHostMalloc(C.byref(hptr), C.c_size_t(n), 0) == 0
assert hip.hipStreamCreateWithFlags(C.byref(s1), 1) == 0
assert hip.hipStreamCreateWithFlags(C.byref(s2), 1) == 0
assert hip.hipEventCreateWithFlags(C.byref(ev), 2) == 0


def run(label, fn):
    for _ in range(2):
        hip.hipDeviceSynchronize()
        t = time.perf_counter()
        fn()
        hip.hipDeviceSynchronize()
        dt = time.perf_counter() - t
    print(f"{label}: {dt * 1e3:.2f} ms", flush=True)
    time.sleep(0.03)


def copy(stream):
    assert hip.hipMemcpyAsync(hptr, dptr, C.c_size_t(n), 2, stream) == 0


run("a idle stream", lambda: copy(s1))
run("b memset on the same stream, then copy", lambda: (hip.hipMemsetAsync(dptr, 0, C.c_size_t(n), s1), copy(s1)))
run("c memset on s2, event, s1 waits, copy on s1",
    lambda: (hip.hipMemsetAsync(dptr, 0, C.c_size_t(n), s2), hip.hipEventRecord(ev, s2), hip.hipStreamWaitEvent(s1, ev, 0), copy(s1)))
run("d device-to-device copy on the same stream, then copy", lambda: (hip.hipMemcpyAsync(d2, dptr, C.c_size_t(n), 3, s1), copy(s1)))
run("e host-to-device copy on the same stream, then copy", lambda: (hip.hipMemcpyAsync(d2, hptr, C.c_size_t(n), 1, s1), copy(s1)))
