#!/usr/bin/env python3
"""Per-base scores to the host: the plain int64 copy against the narrow wire (int16 across PCIe, widened by the host
threads; FTK_WPS_NARROW_WIRE, read when the library is loaded - so one child process per setting).  A 30x contig is
generated on the device, `Engine.wps` of all of it (kernel + copy back into a page-locked result array) is timed.
usage: tools/wps_wire_ab.py [contig=1] [reps=5]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(contig, reps):
    import numpy as np
    import torch
    from finaletoolkit_amd import synth
    from finaletoolkit_amd.engine import Engine
    size = synth.B37_SIZES[contig]
    dev = torch.device("cuda", 0)
    s, e, q, st = synth.gen_contig_device(torch, dev, size, synth.n_fragments(size, 30.0), 7)
    with Engine(0) as eng:
        eng.load_contig_device("c", s, e, q, st, int(s.numel()))
        times, first = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            w = eng.wps("c", 0, size, size)
            times.append(round(time.perf_counter() - t0, 4))
            if first is None:
                first = (int(w.sum()), int(w.min()), int(w.max()), int(np.abs(w[::997]).sum()))
            del w
        dev_out = torch.empty(size, dtype=torch.int64, device=dev)
        eng.wps("c", 0, size, size, out=dev_out)
        eng.sync()
        ref = dev_out.cpu().numpy()
        w = eng.wps("c", 0, size, size)
        same = bool(np.array_equal(w, ref))
    print(json.dumps({"narrow_wire": os.environ.get("FTK_WPS_NARROW_WIRE", "1"), "contig": contig, "bases": size,
                      "GB": round(size * 8 / 1e9, 2), "seconds": times, "equal_to_device_result": same, "digest": first}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        contig = sys.argv[1] if len(sys.argv) > 1 else "1"
        reps = sys.argv[2] if len(sys.argv) > 2 else "5"
        for setting in ("0", "1"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", contig, reps],
                           env=dict(os.environ, FTK_WPS_NARROW_WIRE=setting), check=False)
