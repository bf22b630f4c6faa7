#!/usr/bin/env python3
"""Do kernels slow down while a large device -> page-locked host copy runs on another stream, in a torch process
(its bundled HIP runtime carries such copies with a blit kernel on this box, the system runtime with SDMA:
tools/native/copy_probe.hip)?"""
import time

import torch

dev = torch.device("cuda", 0)
n = 1 << 27  # 1 GiB of int64
src = torch.zeros(n, dtype=torch.int64, device=dev)
work = torch.zeros(n, dtype=torch.int64, device=dev)
pin = torch.empty(n, dtype=torch.int64, pin_memory=True)
side = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def kernels(label, with_copy):
    torch.cuda.synchronize()
    if with_copy:
        with torch.cuda.stream(side):
            pin.copy_(src, non_blocking=True)
    e0.record()
    for _ in range(20):
        work.add_(1)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1):.2f} ms for 20 kernels", flush=True)


for _ in range(2):
    kernels("20 add kernels over 1 GiB alone", False)
    kernels("20 add kernels while 1 GiB goes to the host on another stream", True)
t = time.perf_counter()
pin.copy_(src, non_blocking=True)
torch.cuda.synchronize()
print(f"the copy alone: {(time.perf_counter() - t) * 1e3:.1f} ms")
