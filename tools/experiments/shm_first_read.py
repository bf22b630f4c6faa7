#!/usr/bin/env python3
"""Is the FIRST read of a freshly written /dev/shm file slower than the second?  (bench.py's whole-genome BAM leg: the
first repetition's producer waits 3-4 s for its reads, the later ones 0.7 s.)  Writes N GB with 8 threads, reads it
back twice with 8 threads of os.preadv into reused buffers.  usage: python3 tools/experiments/shm_first_read.py [GB=20]"""
import os
import sys
import threading
import time

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
path = "/dev/shm/ftk_first_read.bin"
n = int(gb * (1 << 30))
chunk = 64 << 20
blob = os.urandom(1 << 20) * 64
fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
try:
    def write_part(t, k):
        a = n * t // k // chunk * chunk
        b = n * (t + 1) // k // chunk * chunk if t + 1 < k else n
        off = a
        while off < b:
            m = min(chunk, b - off)
            os.pwrite(fd, blob[:m], off)
            off += m
    t0 = time.perf_counter()
    ts = [threading.Thread(target=write_part, args=(t, 8)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    print("write %.1f GB: %.2f s" % (gb, time.perf_counter() - t0), flush=True)

    bufs = [bytearray(chunk) for _ in range(8)]
    def read_part(t, k):
        a = n * t // k // chunk * chunk
        b = n * (t + 1) // k // chunk * chunk if t + 1 < k else n
        off = a
        while off < b:
            m = min(chunk, b - off)
            got = os.preadv(fd, [memoryview(bufs[t])[:m]], off)
            off += got
    for r in range(3):
        t0 = time.perf_counter()
        ts = [threading.Thread(target=read_part, args=(t, 8)) for t in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        print("read %d: %.2f s  %.1f GB/s" % (r, dt, gb * 1.0737 / dt), flush=True)
finally:
    os.close(fd)
    os.remove(path)
