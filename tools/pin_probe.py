#!/usr/bin/env python3
"""Cost of page-locking a large result buffer, three ways: hipHostMalloc, hipHostRegister of a fresh 4 KB-page
mapping, hipHostRegister of a mapping that asked for 2 MB pages - and the device -> host copy rate into each
and into plain pageable memory.  usage: tools/pin_probe.py [MB=410]"""
import ctypes as C
import mmap
import sys
import time

import torch

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 410
n = MB << 20
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
src = torch.ones(n // 8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
D2H = 2


def copy_rate(dst):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(dst, src.data_ptr(), n, D2H) == 0
    return time.perf_counter() - t0


def addr(m):
    return C.addressof(C.c_char.from_buffer(m))


for rep in range(2):
    t0 = time.perf_counter()
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), n, 0) == 0
    t_alloc = time.perf_counter() - t0
    t_copy = copy_rate(p)
    t_copy2 = copy_rate(p)
    hip.hipHostFree(p)
    print(f"hipHostMalloc        : alloc {t_alloc*1e3:7.1f} ms  first copy {t_copy*1e3:6.1f} ms  second {t_copy2*1e3:6.1f} ms ({n/t_copy2/1e9:.1f} GB/s)")
    for huge in (False, True):
        t0 = time.perf_counter()
        m = mmap.mmap(-1, n + (2 << 20))
        if huge:
            m.madvise(mmap.MADV_HUGEPAGE)
        a = (addr(m) + (2 << 20) - 1) & ~((2 << 20) - 1)
        t_map = time.perf_counter() - t0
        t0 = time.perf_counter()
        C.memset(a, 0, n)  # fault the pages in (one thread)
        t_touch = time.perf_counter() - t0
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(a, n, 0)
        t_reg = time.perf_counter() - t0
        t_copy = copy_rate(a) if rc == 0 else float("nan")
        t_copy2 = copy_rate(a) if rc == 0 else float("nan")
        if rc == 0:
            hip.hipHostUnregister(a)
        print(f"register {'2MB' if huge else '4KB'} pages   : map {t_map*1e3:5.1f}  touch {t_touch*1e3:6.1f}  register {t_reg*1e3:6.1f} ms (rc {rc})  "
              f"first copy {t_copy*1e3:6.1f} ms  second {t_copy2*1e3:6.1f} ms")
        del a
        m.close()
    for huge in (False, True):
        m = mmap.mmap(-1, n + (2 << 20))
        if huge:
            m.madvise(mmap.MADV_HUGEPAGE)
        a = (addr(m) + (2 << 20) - 1) & ~((2 << 20) - 1)
        t_copy = copy_rate(a)
        t_copy2 = copy_rate(a)
        print(f"pageable {'2MB' if huge else '4KB'} pages   : first copy (faults the pages) {t_copy*1e3:6.1f} ms ({n/t_copy/1e9:.1f} GB/s)  second {t_copy2*1e3:6.1f} ms ({n/t_copy2/1e9:.1f} GB/s)")
        m.close()
