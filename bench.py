#!/usr/bin/env python3
"""
bench.py -- genomic windows/sec (coverage + fragment-length histogram + DELFI +
WPS) at 30x WGS on 1/2/4/8 MI355X.

One "step" = one pass of the whole hot path over every 100 kb window of the
rank's contigs: per-window coverage counts, per-window length histograms
(1001 bins), DELFI short/long counts (blacklist + gap constants) and WPS for
every base, followed by the all-gather of the DELFI bin vector (N > 1).
Fragments are synthetic (seeded, BASELINE.md section 4 distribution), generated
on the device and resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W [--depth 30] [--contigs 1,2,...]

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from finaletoolkit_amd import synth  # noqa: E402
from finaletoolkit_amd.synth import gen_contig_device, synth_blacklist, synth_gaps  # noqa: E402,F401
from finaletoolkit_amd.sharding import launch_ranks, split_units, unit_halo  # noqa: E402

WINDOW = 100_000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HIST_BINS = 1001
WPS_W, WPS_MIN, WPS_MAX, MAPQ = 120, 120, 180, 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--contigs", type=str, default="", help="comma list (default: b37 1-22,X,Y)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the file -> result legs (end_to_end in the JSON)")
    ap.add_argument("--no-kernel-rows", action="store_true", help="skip the next-row / BAM-mode kernel rooflines (tools/kernel_rows.py)")
    args = ap.parse_args()
    if os.environ.get("FTK_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after <seconds>
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FTK_BENCH_WATCHDOG"]), exit=True)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N fresh rank processes
        # (it has not touched the GPU and never will) and exits with their status
        raise SystemExit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                      share_gpu=os.environ.get("FTK_BENCH_SHARE_GPU") == "1"))

    # (multi-process GPU work on this platform needs the dmabuf IPC mode - the host driver supports no other; exported
    # on the build boxes already, kept here for a launcher that starts the ranks from a cleaner environment)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    from finaletoolkit_amd import _lib
    from finaletoolkit_amd.engine import Engine

    _lib.load()  # before the first HIP call of this process: loading sets the decoder's hardware-queue count (FTK_HW_QUEUES)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         f"they must agree (one rank per GPU)")
    share = os.environ.get("FTK_BENCH_SHARE_GPU") == "1"  # test mode: every rank on GPU 0, gloo exchange
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    if not share and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible MI355X devices, found "
                         f"{torch.cuda.device_count()} (one rank per GPU; no fallback to fewer)")
    if share:
        local = 0
        os.environ.setdefault("FTK_BENCH_DIST_BACKEND", "gloo")
        # (the product's own engine - frag.delfi in the N-rank file leg - picks its GPU from FTK_DEVICE, else LOCAL_RANK:
        # under torch.distributed.run LOCAL_RANK is the rank, and rank 1 has no GPU 1 on a shared box - it failed there,
        # alone, and the other ranks waited for it in the leg's first collective)
        os.environ["FTK_DEVICE"] = "0"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # FTK_BENCH_FORCE_DIST=1 drives the collective code path with a 1-rank RCCL communicator (1-GPU boxes)
    use_dist = world > 1 or bool(os.environ.get("FTK_BENCH_FORCE_DIST"))
    # The exchange goes through finaletoolkit_amd/comm.py: "rccl" = the library's own communicator (ftk_comm_* of
    # include/ftk.h: RCCL over xGMI, device buffers, no torch.distributed); FTK_BENCH_DIST_BACKEND=gloo (test mode):
    # torch.distributed over host copies, so several ranks can share one GPU (LOCAL_RANK=0 for all) and the
    # multi-rank logic can be exercised on a 1-GPU box
    backend = os.environ.get("FTK_BENCH_DIST_BACKEND", "rccl")
    grp = None
    if use_dist:
        from finaletoolkit_amd import comm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # (ten minutes instead of the backends' 10-30: a rank that failed alone must not hold the others for half an hour)
        os.environ.setdefault("FTK_DIST_TIMEOUT_S", "600")
        os.environ["FTK_DEVICE"] = str(local)
        if world > 1:
            grp = comm.join(backend)
        else:
            grp = comm._GROUP = comm.RcclGroup(0, 1, local, None)

    sizes = dict(synth.B37_SIZES)
    if args.contigs:
        sizes = {k: sizes[k] for k in args.contigs.split(",")}
    names = list(sizes)
    # Work units: whole contigs on one GPU; with N GPUs the genome is cut into N equal runs at window
    # boundaries (a rank owns whole contigs plus at most two partial ones, sharding.split_units).  A
    # partial unit holds the contig's fragments that start within `halo` of its range and computes
    # exactly its own windows / bases, so there is still no data-path exchange.
    units = split_units(sizes, world, WINDOW)
    sim = os.environ.get("FTK_BENCH_SIM")  # "world:rank": run that rank's units of a world-GPU split alone
    if sim:
        sim_world, sim_rank = (int(x) for x in sim.split(":"))
        units = [(0, c, a, b) for r, c, a, b in split_units(sizes, sim_world, WINDOW) if r == sim_rank]
    ukey = lambda u: f"{u[1]}:{u[2]}-{u[3]}"
    mine = [u for u in units if u[0] == rank]
    halo = unit_halo(1000, WPS_W)

    eng = Engine(local)
    # one explicit stream for torch ops, RCCL and the ftk launches (a NULL handle would mean "ctx's own stream")
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    if not os.environ.get("FTK_BENCH_OWN_STREAM"):
        eng.set_stream(stream.cuda_stream)

    # ---- resident inputs (untimed) -------------------------------------------
    t_load = time.time()
    per = {}
    cached = (None, None)
    # One buffer per output for the whole rank (its units back to back): the batched launches write rows
    # / bases by global index, the per-unit calls get views.
    rows_rank = sum(-(-u[3] // WINDOW) - u[2] // WINDOW for u in mine)
    pad32 = lambda n: (n + 31) // 32 * 32  # every unit's scores start 256-byte aligned (16-byte vector stores)
    bases_rank = sum(pad32(u[3] - u[2]) for u in mine)
    all_cov = torch.zeros(rows_rank, dtype=torch.int64, device=dev)
    all_hist = torch.zeros((rows_rank, HIST_BINS), dtype=torch.int32, device=dev)
    all_over = torch.zeros(rows_rank, dtype=torch.int64, device=dev)
    all_wps = torch.empty(bases_rank, dtype=torch.int64, device=dev)
    row0 = base0 = 0
    # the unit the CPU baselines sample: the rank's largest (chr1 at N = 1: 2 493 windows - the reference-shaped Python
    # sample needs 2 000)
    sample_unit = max(mine, key=lambda u: u[3] - u[2]) if mine else None
    for u in mine:
        _, c, a, b = u
        ci = names.index(c)
        if cached[0] != c:
            n_c = synth.n_fragments(sizes[c], args.depth)
            cached = (None, None)
            cached = (c, gen_contig_device(torch, dev, sizes[c], n_c, synth.SEED_BASE + ci))
        s, e, q, st = cached[1]
        if (a, b) != (0, sizes[c]):
            lo = int(torch.searchsorted(s, torch.tensor([a - halo], dtype=torch.int32, device=dev)).item())
            hi = int(torch.searchsorted(s, torch.tensor([b + halo], dtype=torch.int32, device=dev)).item())
            s, e, q, st = (t[lo:hi].contiguous() for t in (s, e, q, st))
        n = int(s.numel())
        torch.cuda.synchronize()
        eng.load_contig_device(ukey(u), s, e, q, st, n)
        ws, we = synth.tiling_windows(sizes[c], WINDOW)
        inside = (ws >= a) & (ws < b)
        ws, we = np.ascontiguousarray(ws[inside]), np.ascontiguousarray(we[inside])
        bl_s, bl_e = synth_blacklist(sizes[c], 77 + ci, max(8, int(2000 * sizes[c] / 3.1e9)))
        nw = len(ws)
        per[ukey(u)] = dict(
            contig=c, a=a, b=b, size=sizes[c],
            n=n, ws=ws, we=we, nw=nw, bl=(bl_s, bl_e), gaps=synth_gaps(sizes[c]),
            d_ws=torch.from_numpy(ws).to(dev), d_we=torch.from_numpy(we).to(dev),
            cov=all_cov[row0:row0 + nw], hist=all_hist[row0:row0 + nw], over=all_over[row0:row0 + nw],
            wps=all_wps[base0:base0 + (b - a)], wps_off=base0,
            keep=(s, e, q, st) if u == sample_unit else None,
        )
        row0 += nw
        base0 += pad32(b - a)
        del s, e, q, st
    cached = (None, None)
    mine = [ukey(u) for u in mine]
    torch.cuda.synchronize()
    t_load = time.time() - t_load

    import ctypes as C
    from finaletoolkit_amd import _lib as L
    lib = eng.lib
    flt = L.make_filter(MAPQ, None, None, "midpoint")
    for c in mine:
        per[c]["gaps_c"] = L.make_gaps(per[c]["gaps"])
    unit_rows = lambda u: -(-u[3] // WINDOW) - u[2] // WINDOW
    bins_all = sum(unit_rows(u) for u in units)
    max_bins_rank = max(sum(unit_rows(u) for u in units if u[0] == r) for r in range(world))
    # The DELFI (short, long) vectors are written by the kernels straight into the all-gather send
    # buffer: row 0 = short, row 1 = long, this rank's contigs back to back (no packing kernels).
    gather_in = torch.zeros((2, max_bins_rank), dtype=torch.int64, device=dev)
    native = use_dist and grp.backend == "rccl"
    # rank r's rows land in gather_out[r]: in HBM for the library's communicator, on the host in the gloo test mode
    gather_out = torch.zeros((world, 2, max_bins_rank), dtype=torch.int64, device=dev if native else "cpu") if use_dist else None
    if native:
        grp.set_stream(stream.cuda_stream)  # the collectives fork from, and join into, the stream of the launches
    collect = [True]  # priming steps skip the collective: their count differs from rank to rank

    # The all-gather only needs the DELFI rows, which are complete once the step's LAST window-feature launch
    # is on the stream: it is started there (the communicator's own stream, ordered after the feature pass) and runs
    # concurrently with the WPS launches that follow; the step ends by making the compute stream wait for it.
    def exchange_start():
        if not (use_dist and collect[0]):
            return None
        if not native:  # test backend: host copies, synchronous
            gather_out.copy_(torch.from_numpy(grp.all_gather_i64(gather_in.cpu().numpy())).reshape(gather_out.shape))
            return None
        grp.all_gather_i64_device(gather_in, gather_in.numel(), gather_out)
        return grp

    def exchange_finish(work):
        if work is not None:
            work.join()
    r0 = 0
    for c in mine:
        per[c]["short"] = gather_in[0, r0:r0 + per[c]["nw"]]
        per[c]["long"] = gather_in[1, r0:r0 + per[c]["nw"]]
        r0 += per[c]["nw"]
    wps_ev = {}
    per_rank_order = {r: [ukey(u) for u in units if u[0] == r] for r in range(world)}
    per_rank_rows = {r: [unit_rows(u) for u in units if u[0] == r] for r in range(world)}

    # Launch shape.  Per unit (default for large units): features(c) then WPS(c), so WPS finds the contig's
    # columns in the Infinity Cache.  Batched (default when the rank's units average < 70 Mb, i.e. a rank
    # holding several small contigs): ONE window-feature launch and ONE WPS launch for all of the rank's
    # units (ftk_window_features_batch / ftk_wps_batch) -- better occupancy than 500-window launches.
    # Measured on simulated ranks: 8-GPU rank 7 (6 units) 0.717 -> 0.681 ms, 8-GPU rank 0 (2 units)
    # 0.656 -> 0.667, whole genome on one GPU 5.26 -> 5.46.  FTK_BENCH_BATCH = 0 / 1 (features only) / 2 forces.
    mode = os.environ.get("FTK_BENCH_BATCH", "auto")
    if mode == "auto":
        mode = "2" if mine and sum(per[c]["b"] - per[c]["a"] for c in mine) / len(mine) < 70e6 else "0"
    batched = mode != "0"
    fbatch = eng.feature_batch([dict(name=c, starts=per[c]["ws"], stops=per[c]["we"], bl_start=per[c]["bl"][0],
                                     bl_end=per[c]["bl"][1], gaps=per[c]["gaps"]) for c in mine], MAPQ) if batched else None
    wps_names = list(mine)
    wps_a = [per[c]["a"] for c in mine]
    wps_b = [per[c]["b"] for c in mine]
    wps_cs = [per[c]["size"] for c in mine]
    wps_off = [per[c]["wps_off"] for c in mine]

    # mode 1 keeps WPS one launch per unit (the batched WPS kernel fetches its item -- contig view, interval --
    # from memory at block start: 4.84 vs 4.25 ms for the whole genome).
    batch_wps = mode == "2"

    def step_batched(record_events=False):
        eng.window_features_batch(fbatch, coverage=all_cov, hist=all_hist, hist_bins=(0, HIST_BINS), overflow=all_over,
                                  delfi_q=MAPQ, short=gather_in[0], long=gather_in[1])
        work = exchange_start()
        if batch_wps:
            if record_events:
                eng.event_record(0)
            eng.wps_batch(wps_names, wps_a, wps_b, wps_cs, wps_off, all_wps, WPS_W, WPS_MIN, WPS_MAX, MAPQ)
            if record_events:
                eng.event_record(1)
                wps_ev["batch"] = (0, 1)
        else:
            ev = 0
            for c in mine:
                p = per[c]
                if record_events:
                    eng.event_record(ev)
                eng.wps(c, p["a"], p["b"], p["size"], WPS_W, WPS_MIN, WPS_MAX, MAPQ, out=p["wps"])
                if record_events:
                    eng.event_record(ev + 1)
                    wps_ev[c] = (ev, ev + 1)
                    ev += 2
        exchange_finish(work)

    # FTK_BENCH_FUSED=1 (experiment / BASELINE config 5 shape): ONE launch per whole-contig unit computes WPS and
    # the window features together (ftk_wps_window_features), reading the fragment columns once.
    fused = os.environ.get("FTK_BENCH_FUSED", "0") == "1" and all(per[c]["a"] == 0 and per[c]["b"] == per[c]["size"]
                                                                   for c in mine)

    def step_fused(record_events=False):
        ev = 0
        for c in mine:
            p = per[c]
            if record_events:
                eng.event_record(ev)
            eng._check(lib.ftk_wps_window_features(
                eng.ctx, eng.contig_id(c), 0, p["size"], p["size"], WPS_W, WPS_MIN, WPS_MAX, MAPQ, L.ptr(p["wps"]), 0,
                WINDOW, p["nw"], C.byref(flt), L.ptr(p["cov"]), 0, HIST_BINS, L.ptr(p["hist"]), L.ptr(p["over"]), MAPQ,
                L.ptr(p["bl"][0]), L.ptr(p["bl"][1]), len(p["bl"][0]), C.byref(p["gaps_c"]), L.ptr(p["short"]),
                L.ptr(p["long"])))
            if record_events:
                eng.event_record(ev + 1)
                wps_ev[c] = (ev, ev + 1)
                ev += 2
        exchange_finish(exchange_start())

    # FTK_BENCH_MERGED (default 1): ONE launch per unit whose grid holds the window-feature blocks first and the WPS
    # tiles behind them (ftk_window_features_wps / feat_then_wps_kernel): the feature pass's tail and the WPS ramp
    # overlap instead of adding up.  0 = the two launches per unit of rounds 1-2.
    merged = os.environ.get("FTK_BENCH_MERGED", "1") != "0"

    def step_merged(record_events=False):
        ev = 0
        for c in mine:
            p = per[c]
            if record_events:
                eng.event_record(ev)
            eng._check(lib.ftk_window_features_wps(
                eng.ctx, eng.contig_id(c), L.ptr(p["ws"]), L.ptr(p["we"]), p["nw"], C.byref(flt), L.ptr(p["cov"]), 0,
                HIST_BINS, L.ptr(p["hist"]), L.ptr(p["over"]), MAPQ, L.ptr(p["bl"][0]), L.ptr(p["bl"][1]),
                len(p["bl"][0]), C.byref(p["gaps_c"]), L.ptr(p["short"]), L.ptr(p["long"]), p["a"], p["b"], p["size"],
                WPS_W, WPS_MIN, WPS_MAX, MAPQ, L.ptr(p["wps"])))
            if record_events:
                eng.event_record(ev + 1)
                wps_ev[c] = (ev, ev + 1)
                ev += 2
        exchange_finish(exchange_start())

    def step(record_events=False):
        if fused:
            return step_fused(record_events)
        if batched:
            return step_batched(record_events)
        if merged:
            return step_merged(record_events)
        row = 0
        ev = 0
        work = None
        split = bool(os.environ.get("FTK_BENCH_SPLIT_ORDER"))  # experiment: all feature passes, then all WPS
        if record_events:
            eng.event_record(4000)  # start of the step: the first unit's feature pass begins here
        for phase in ((0, 1) if split else (2,)):
            for c in mine:
                p = per[c]
                cid = eng.contig_id(c)
                if phase in (0, 2):
                    # coverage + length histogram + DELFI short/long in ONE pass over the contig's fragments.
                    # Windows/blacklist are host arrays: hashed, their device form (windows + per-window
                    # blacklist CSR) is cached in the ctx; every output stays on the device (no sync).
                    eng._check(lib.ftk_window_features(
                        eng.ctx, cid, L.ptr(p["ws"]), L.ptr(p["we"]), p["nw"], C.byref(flt), L.ptr(p["cov"]), 0,
                        HIST_BINS, L.ptr(p["hist"]), L.ptr(p["over"]), MAPQ, L.ptr(p["bl"][0]), L.ptr(p["bl"][1]),
                        len(p["bl"][0]), C.byref(p["gaps_c"]), L.ptr(p["short"]), L.ptr(p["long"])))
                    if c == mine[-1]:  # the rank's DELFI rows are complete: start the exchange behind it
                        work = exchange_start()
                if phase in (1, 2):
                    if record_events:
                        eng.event_record(ev)
                    eng.wps(c, p["a"], p["b"], p["size"], WPS_W, WPS_MIN, WPS_MAX, MAPQ, out=p["wps"])
                    if record_events:
                        eng.event_record(ev + 1)
                        wps_ev[c] = (ev, ev + 1)
                        ev += 2
        exchange_finish(work)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            grp.barrier()
        torch.cuda.synchronize()

    # The host enqueues a step (about 100 asynchronous launches) much faster than the GPU runs it; keep at
    # most two steps in flight (wait for step i-2 before enqueuing step i) so the queue depth does not
    # grow with --steps.
    throttle = os.environ.get("FTK_BENCH_THROTTLE", "1") != "0"
    done_ev = []

    trace = [] if os.environ.get("FTK_BENCH_TRACE") else None

    def run_steps(k, timed):
        for i in range(k):
            if throttle and len(done_ev) >= 2:
                done_ev[-2].synchronize()
            if trace is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                h0 = time.perf_counter()
            # the last step carries the per-launch HIP events of the roofline; the last warm-up step does
            # too, so that the lazily created event slots exist before the timed region starts
            step(record_events=(i == k - 1))
            if trace is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(stream)
                trace.append((timed, time.perf_counter() - h0, e0, e1))
            if throttle:
                ev = torch.cuda.Event()
                ev.record(stream)
                done_ev.append(ev)
                del done_ev[:-2]

    # One-time runtime effect, measured on this stack (ROCm 7.2): the ~830th kernel launch of a process stalls
    # host and GPU for 35-50 ms (seen at step 8 of the 24-contig run and at step 52 of a 4-contig run, never
    # again in 5 000+ launches).  Setup therefore primes the runtime with untimed steps until that many launches
    # are behind us, so the stall cannot land in a short warm-up + timed region.  FTK_BENCH_PRIME=0 turns it off.
    launches_per_step = (3 if batch_wps else 2 + len(mine)) if batched else (2 if merged else 4) * len(mine) + 1
    prime = 0
    if os.environ.get("FTK_BENCH_PRIME", "1") != "0":
        prime = min(512, -(-1200 // launches_per_step))
        collect[0] = False
        run_steps(prime, False)
        collect[0] = True
        barrier()
        done_ev.clear()
    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:  # the slowest rank's time
        dt = max(grp.all_gather_object(dt))
    ms_per_step = dt * 1e3 / args.steps
    if trace is not None and rank == 0:
        for timed, host_s, e0, e1 in trace:
            sys.stderr.write(f"step timed={int(timed)} host_enqueue={host_s * 1e3:7.3f} ms  gpu={e0.elapsed_time(e1):7.3f} ms\n")
    value = bins_all * args.steps / dt

    # ---- roofline of the dominant kernel (WPS), from the last timed step -------
    wps_bytes = 0
    wps_ms = 0.0
    one_launch = merged and not batched and not fused  # the events bracket feature blocks + WPS tiles of a unit
    feat_unit_bytes = lambda u: 10 * per[u]["n"] + 8 * per[u]["nw"] + per[u]["nw"] * (4 * HIST_BINS + 8 * 4)
    for c, (a, b) in wps_ev.items():
        wps_ms += eng.event_elapsed_ms(a, b)
        for u in (mine if c == "batch" else [c]):
            wps_bytes += 10 * per[u]["n"] + 8 * (per[u]["b"] - per[u]["a"])
            if one_launch:
                wps_bytes += feat_unit_bytes(u)
    achieved = wps_bytes / (wps_ms * 1e-3) / 1e9 if wps_ms > 0 else 0.0
    traffic = traffic_source = None  # HBM bytes per launch from the committed PMC passes (same workload only)
    tpath = os.path.join(ROOT, "profiles", "wps_traffic.json")
    if world == 1 and not sim and not args.contigs and args.depth == 30.0 and os.path.exists(tpath):
        tj = json.load(open(tpath))  # measured per step (the dominant kernel's launches of one step), reported per launch
        if tj.get("kernel", "wps_stream_kernel") == ("feat_then_wps_kernel" if one_launch else "wps_stream_kernel"):
            traffic = int(tj["hbm_bytes_per_step"] / max(len(wps_ev), 1))
            # a committed measurement of this workload (two separate --pmc passes under rocprofv3: tools/profile_round.sh),
            # not a counter read of THIS run: the line says which build's
            traffic_source = f"{tj.get('source', 'profiles/wps_traffic.json')} ({tj.get('build', 'build not recorded')})"
    # second kernel of the step: the fused window-feature pass of a unit runs between the previous unit's WPS
    # stop event and this unit's WPS start event (per-unit launch shape only)
    feat = None
    if not batched and not fused and not one_launch and wps_ev and not os.environ.get("FTK_BENCH_SPLIT_ORDER"):
        f_ms, f_bytes, prev = 0.0, 0, 4000
        for c in mine:
            a, b = wps_ev[c]
            f_ms += eng.event_elapsed_ms(prev, a)
            f_bytes += 10 * per[c]["n"] + 8 * per[c]["nw"] + per[c]["nw"] * (4 * HIST_BINS + 8 * 4)
            prev = b
        feat = dict(kernel="feat_fast_kernel" if os.environ.get("FTK_FEAT_FAST", "1") != "0" else "feat_block_kernel", achieved=round(f_bytes / (f_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(f_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    algorithmic_bytes_per_launch=int(f_bytes / len(mine)), launches=len(mine),
                    avg_launch_ms=round(f_ms / len(mine), 4),
                    note="follows a WPS launch: reads behind that kernel's write-back")
    if os.environ.get("FTK_BENCH_DETAIL") and rank == 0 and not batched and not fused and not one_launch:
        prev = 4000
        for c in mine:  # per unit: fragments, feature-pass and WPS launch durations of the last timed step
            a, b = wps_ev[c]
            f_us, w_us = eng.event_elapsed_ms(prev, a) * 1e3, eng.event_elapsed_ms(a, b) * 1e3
            fb = 10 * per[c]["n"] + per[c]["nw"] * (4 * HIST_BINS + 40)
            wb = 10 * per[c]["n"] + 8 * (per[c]["b"] - per[c]["a"])
            sys.stderr.write(f"unit {c:>24s} frags {per[c]['n']:>9d} feat {f_us:7.1f} us {fb / f_us / 1e6:6.2f} TB/s   "
                             f"wps {w_us:7.1f} us {wb / w_us / 1e6:6.2f} TB/s\n")
            prev = b
    roofline = dict(bound="hbm", kernel="feat_then_wps_kernel" if one_launch else "wps_stream_kernel",
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                    algorithmic_bytes_per_launch=int(wps_bytes / max(len(wps_ev), 1)),
                    launches=len(wps_ev), avg_launch_ms=round(wps_ms / max(len(wps_ev), 1), 4))
    if feat:
        roofline["second_kernel"] = feat
    # the whole step against the same roof: algorithmic bytes of every launch of this rank / wall time per step
    step_bytes = sum(2 * 10 * per[c]["n"] + 8 * (per[c]["b"] - per[c]["a"]) + 8 * per[c]["nw"]
                     + per[c]["nw"] * (4 * HIST_BINS + 8 * 4) for c in mine)
    roofline["whole_step"] = dict(algorithmic_bytes=int(step_bytes), achieved=round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                  unit="GB/s", frac=round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))

    # ---- size-independent checks on the full workload ---------------------------
    checks = {}
    tot_cov = sum(int(per[c]["cov"].sum().item()) for c in mine)
    tot_hist = sum(int(per[c]["hist"].sum().item()) + int(per[c]["over"].sum().item()) for c in mine)
    checks["cov_sum_eq_hist_sum"] = tot_cov == tot_hist

    if use_dist:  # every rank must hold every contig's (short, long) rows, in LPT rank order
        torch.cuda.synchronize()
        ok = True
        for c in mine:
            o = sum(per_rank_rows[rank][:per_rank_order[rank].index(c)])
            ok = ok and torch.equal(gather_out[rank, 0, o:o + per[c]["nw"]].cpu(), per[c]["short"].cpu()) \
                and torch.equal(gather_out[rank, 1, o:o + per[c]["nw"]].cpu(), per[c]["long"].cpu())
        checks["allgather_roundtrip"] = bool(ok)
        # every rank now holds the whole-genome vector: total DELFI fragments must match on all ranks
        tot = int(gather_out.sum().item())
        checks["allgather_same_on_all_ranks"] = bool(all(t == tot for t in grp.all_gather_object(tot)))
        # and it must be the whole genome's DELFI count: compare with the sum of what every rank computed itself
        mine_tot = int(grp.all_reduce_sum_i64(np.array([int(gather_in.sum().item())], np.int64))[0])
        checks["allgather_total_eq_sum_of_ranks"] = mine_tot == tot
    # ---- the kernels behind the headline launch: the section-8(f) rows and the BAM-mode kernels of config 5 ----------
    next_rows = None
    if rank == 0 and world == 1 and not sim and not args.no_kernel_rows:
        try:
            from tools import kernel_rows
            kr = kernel_rows.measure(torch, eng, "all", reps=5)
            next_rows = kr.get("next_rows")
            roofline["bam_kernels"] = kr.get("bam_kernels")
        except Exception as exc:  # noqa: BLE001 - the headline line must still be printed
            next_rows = {"error": f"{type(exc).__name__}: {exc}"}
    file_leg = None
    if use_dist and world > 1 and not sim and not args.no_end_to_end:
        # N ranks, ONE file: BASELINE config 4 end to end through the product function (every rank index-seeks and
        # decodes its own contigs, counts them, one all-gather, rank 0 merges and writes).  The resident workload of
        # the timed steps is released first, on every rank.
        keep_last = per[mine[-1]]["keep"] if mine else None
        for c in list(per):
            eng.release(c)
            for k in ("cov", "hist", "over", "wps", "short", "long", "d_ws", "d_we", "keep"):
                per[c].pop(k, None)
        del all_wps, all_hist, all_cov, all_over
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        file_leg = multi_rank_file_leg(torch, grp, rank, world, {c: sizes[c] for c in names}, args.depth)
    e2e = None
    if rank == 0 and world == 1 and not sim and not args.no_end_to_end:
        # file -> result legs (SURVEY 8-d's second figure).  The resident workload is released first.
        keep_last = per[mine[-1]]
        cpu_inputs = dict(per=per, mine=mine) if not args.no_cpu_baseline else None
        e2e = "pending"
    out = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            cpu = cpu_baseline(torch, eng, per, ukey(sample_unit), args.cpu_seconds, checks)
        if e2e == "pending":
            for c in list(per):
                eng.release(c)
            del all_wps, all_hist, all_cov, all_over, gather_in
            for c in per:
                for k in ("cov", "hist", "over", "wps", "short", "long", "d_ws", "d_we", "keep"):
                    per[c].pop(k, None)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            e2e = end_to_end(torch, cpu=cpu)
        out = {
            "metric": "genomic windows/sec (coverage+WPS+DELFI) at 30x WGS",
            "value": round(value, 1), "unit": "windows/s", "n_gpus": world,
            "rccl_ranks": (grp.world if native else None),  # the size ftk_comm_size reports for the library's communicator
            "exchange": (f"{grp.backend} all-gather ({'libftk_hip.so ftk_comm_*: RCCL' if native else 'torch.distributed, test mode'}), "
                         f"{grp.world} ranks" if use_dist else "none (1 rank)"),
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int32/int64", "data": "synthetic",
            "config": {"workload": f"whole-genome b37 1-22,X,Y synthetic {args.depth:g}x fragments "
                                   f"({sum(synth.n_fragments(sizes[c], args.depth) for c in names)} fragments), "
                                   f"{bins_all} x 100 kb windows: coverage + length histogram (1001 bins) + DELFI "
                                   f"short/long (blacklist+gaps) + WPS W=120 every base",
                       "contigs": len(names), "windows": bins_all, "depth": args.depth,
                       "sharding": (f"genome cut into {world} equal window-aligned runs ({len(units)} units, halo "
                                    f"{halo} bp); all-gather of DELFI bin vector") if world > 1
                       else ("single GPU" if not sim else f"simulated rank {sim} alone")},
            "roofline": roofline, "next_rows": next_rows, "cpu_baseline": cpu, "end_to_end": e2e if world == 1 else file_leg, "checks": checks, "load_s": round(t_load, 2),
            "priming_steps": prime, "launches": "fused WPS + features, 1 launch per unit" if fused else "1 launch per unit: feature blocks, then WPS tiles" if one_launch else ("1 batched feature launch + " + ("1 batched WPS launch" if batch_wps else "1 WPS launch per unit")
                         + " per step") if batched else "per unit",
        }
    if use_dist:
        comm.leave()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush that first so the JSON line is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def rep_summary(runs):
    """A leg's repetitions, in order: the best one's stage split, plus ``first_s`` (what a process's first call pays:
    thread-pool start, page-locked allocations, the file's first read), ``median_s`` and ``best_s``.  ``total_s`` stays
    the best repetition (the figure of rounds 1-3); ``results_ok`` is true only if every repetition's check passed."""
    times = [r["total_s"] for r in runs]
    out = dict(min(runs, key=lambda r: r["total_s"]))
    out.update(first_s=round(times[0], 4), median_s=round(float(np.median(times)), 4), best_s=round(min(times), 4),
               repetitions=len(times), all_s=[round(t, 4) for t in times],
               results_ok=bool(all(r.get("results_ok", True) for r in runs)))
    if runs[0].get("decoder_producer_stage_ms"):  # what the FIRST repetition's producer spent where (a file's first read ...)
        out["first_repetition_producer_stage_ms"] = runs[0]["decoder_producer_stage_ms"]
    if os.environ.get("FTK_BENCH_REP_STAGES") == "1":  # debugging aid: every repetition's stage split, not only the best one's
        out["all_runs"] = [dict(r) for r in runs]
    return out


def measure_h2d_gbs(torch, dev, mb: int = 256) -> float:
    """Host -> device rate of this box for one large page-locked copy (GB/s, best of four): the PCIe term of the file
    legs' floor."""
    src = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    del src, dst
    return (mb << 20) / best / 1e9


def inflate_alone_rates():
    """DEFLATE-alone rates of the device inflate kernel on a chip-filling launch (GB/s of inflated bytes), measured
    with ``tools/inflate_variants.sh`` under rocprofv3 and committed as ``profiles/inflate_alone.json``."""
    path = os.path.join(ROOT, "profiles", "inflate_alone.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


def measure_d2h_gbs(torch, dev, mb: int = 256) -> float:
    """Device -> host rate of this box for one large page-locked copy (GB/s, best of four): the score term of the floor
    of the legs that return per-base results."""
    dst = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    src = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    del src, dst
    return (mb << 20) / best / 1e9


def read_once(path: str, threads: int = 8) -> float:
    """(Untimed preparation of a file leg.)  The file a leg reads was written a moment ago, and the FIRST read of a
    freshly written file is a property of the page cache, not of the decoder: on the box's /dev/shm 17.7 GB/s against
    57 GB/s for every later read (tools/experiments/shm_first_read.py) - three of the 3.4-4.5 s a whole-genome BAM leg's
    first repetition took.  The legs' premise is a file that sits in the page cache, so it is read through once here;
    returns the seconds that took."""
    import threading
    t0 = time.perf_counter()
    try:
        n = os.path.getsize(path)
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return 0.0
    try:
        chunk = 32 << 20
        k = max(1, min(threads, n // chunk + 1))

        def part(t):
            buf = bytearray(chunk)
            off, end = n * t // k, n * (t + 1) // k
            while off < end:
                got = os.preadv(fd, [memoryview(buf)[:min(chunk, end - off)]], off)
                if got <= 0:
                    return
                off += got
        ts = [threading.Thread(target=part, args=(t,)) for t in range(k)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        os.close(fd)
    return time.perf_counter() - t0


def measure_host_write_gbs(threads: int, mb_per_thread: int = 192) -> float:
    """What this box's cores can WRITE to memory together (GB/s): `threads` threads each filling their own array, best of
    three.  An upper bound on any widening of narrow per-base scores into the caller's int64 array, whose result bytes
    must at least be written once - the host term of the floor of the legs that return per-base results."""
    from concurrent.futures import ThreadPoolExecutor
    n = max(1, min(threads, 64))
    bufs = [np.empty(mb_per_thread << 20, np.uint8) for _ in range(n)]
    for b in bufs:
        b.fill(1)  # (pages faulted in before the clock starts)
    best = 1e9
    with ThreadPoolExecutor(n) as ex:
        for _ in range(3):
            t0 = time.perf_counter()
            list(ex.map(lambda b: b.fill(0), bufs))
            best = min(best, time.perf_counter() - t0)
    return n * (mb_per_thread << 20) / best / 1e9


LINK = {}  # measured once per run by end_to_end: d2h_GBps, host_write_GBps


def leg_floor(leg, file_bytes, inflated_bytes, kind, h2d_gbs, rates, score_bases: int = 0):
    """The file legs' own roofline.  No run of the leg can be shorter than (the stages overlap, so the floor is the
    LONGEST of them): the compressed bytes crossing PCIe at the measured H2D rate; the inflate kernel alone on all of its
    blocks; and, for a leg that hands `score_bases` per-base int64 scores to the host, those scores crossing the link the
    other way on the narrow wire (2 B per base at the measured D2H rate) and their 8 B per base being written into the
    caller's array by the host cores (at what the cores can write together, measured)."""
    pcie_s = file_bytes / (h2d_gbs * 1e9) if h2d_gbs else None
    rate = (rates or {}).get({"bam": "bam_GBps", "text": "text_GBps"}.get(kind, kind + "_GBps"))
    infl_s = inflated_bytes / (rate * 1e9) if rate else None
    d2h_s = widen_s = None
    if score_bases and LINK.get("d2h_GBps"):
        d2h_s = 2 * score_bases / (LINK["d2h_GBps"] * 1e9)
    if score_bases and LINK.get("host_write_GBps"):
        widen_s = 8 * score_bases / (LINK["host_write_GBps"] * 1e9)
    terms = [t for t in (pcie_s, infl_s, d2h_s, widen_s) if t]
    if not terms:
        return
    floor = max(terms)
    r4 = lambda v: None if v is None else round(v, 4)  # noqa: E731
    leg["floor"] = dict(floor_s=round(floor, 4), pcie_s=r4(pcie_s), inflate_alone_s=r4(infl_s),
                        h2d_GBps_measured=None if not h2d_gbs else round(h2d_gbs, 1), inflate_GBps=rate,
                        inflate_rate_source=(rates or {}).get("source"),
                        frac_of_floor_best=round(floor / leg["best_s"], 3), frac_of_floor_median=round(floor / leg["median_s"], 3))
    if score_bases:
        leg["floor"].update(d2h_s=r4(d2h_s), host_widen_s=r4(widen_s), score_bases=int(score_bases),
                            d2h_GBps_measured=LINK.get("d2h_GBps"), host_write_GBps_measured=LINK.get("host_write_GBps"))


def multi_rank_file_leg(torch, grp, rank, world, sizes, depth, reps: int = 2):
    """BASELINE config 4 with N ranks: rank 0 writes ONE indexed fragment file of the run's contigs (not timed), then
    every rank calls the product's ``frag.delfi`` on it under the bench's process group: contigs dealt by LPT, a rank
    reads its contigs' blocks through the tabix index and inflates / parses / counts them on ITS GPU with its share
    of the host cores, one all-gather of the bin vectors, rank 0 merges to 5 Mb and writes the TSV.  Reported: the
    slowest rank's wall time (best of ``reps``), every rank's stage split, and whether all ranks hold the same frame."""
    import hashlib
    import shutil
    import tempfile
    import warnings
    from finaletoolkit_amd import bgzf, frag, source, writers
    from finaletoolkit_amd.frag import _delfi as FD
    box = [None]
    t_write = 0.0
    rows_total = 0
    try:
        if rank == 0:
            tmp = tempfile.mkdtemp(prefix="ftk_bench_ranks_")
            dev = torch.device("cuda", torch.cuda.current_device())
            t0 = time.perf_counter()
            pg = os.path.join(tmp, "genome.frag.gz")
            spans, linear = [], []
            names = list(sizes)
            for k, c in enumerate(names):
                n = synth.n_fragments(sizes[c], depth)
                s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, sizes[c], n, synth.SEED_BASE + k))
                with writers.frag_rows(c, s, e, q, st) as text:
                    offs = writers.bgzf_write(pg, text, 1, append=k > 0, write_eof=k == len(names) - 1)
                spans.append((c, int(offs[0]) << 16, int(offs[-1]) << 16))
                linear.append(bgzf.linear_index(s, e, bgzf.row_lengths(c, s, e, q), offs))  # (a rank may own part of a contig)
                rows_total += n
                del s, e, q, st
            bgzf.write_index(pg + ".tbi", False, spans, linear)
            synth.write_genome_delfi_inputs(tmp, sizes, WINDOW)
            synth.write_random_2bit(os.path.join(tmp, "genome.2bit"), sizes)
            t_write = time.perf_counter() - t0
            box[0] = (tmp, rows_total)
        box[0] = grp.broadcast_object(box[0], src=0)
        tmp, rows_total = box[0]
        pg = os.path.join(tmp, "genome.frag.gz")
        side = [os.path.join(tmp, f) for f in ("genome.chrom.sizes", "bins.bed", "blacklist.bed", "gaps.bed", "genome.2bit")]
        n_win = sum(-(-n // WINDOW) for n in sizes.values())
        threads = source.usable_cores()
        best = None
        for _ in range(reps):
            source.close_all()
            del source.REGION_READS[:]
            torch.cuda.synchronize()
            grp.barrier()
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                df = frag.delfi(pg, side[0], side[1], side[4], blacklist_file=side[2], gap_file=side[3],
                                output_file=os.path.join(tmp, "delfi_5mb.tsv"), no_gc_correct=True, merge_bins=True,
                                workers=threads)
            mine_s = time.perf_counter() - t0
            decoded = sorted(k.split(":", 1)[1] for k in source.get_engine().contigs)
            said = grp.all_gather_object(dict(rank=rank, total_s=round(mine_s, 4), stages_s=dict(FD.LAST_STAGE_S),
                                              contigs_decoded=len(decoded), decoder_threads=threads,
                                              regions_read=[list(r[1:]) for r in source.REGION_READS],
                                              frame_sha=hashlib.sha256(df.to_csv(index=False).encode()).hexdigest(),
                                              rows=int(df.shape[0])))
            slowest = max(x["total_s"] for x in said)
            if best is None or slowest < best["total_s"]:
                same = len({x["frame_sha"] for x in said}) == 1
                # whole contigs decoded by one rank each; the contigs a cut falls into read as regions by their owners
                cut = {r[0] for x in said for r in x["regions_read"]}
                parted = sum(x["contigs_decoded"] for x in said) + len(cut) == len(sizes)
                best = dict(total_s=slowest, windows=n_win, windows_per_s=round(n_win / slowest, 1),
                            fragments_per_s_M=round(rows_total / slowest / 1e6, 1), merged_rows=said[0]["rows"],
                            per_rank=[{k: v for k, v in x.items() if k != "frame_sha"} for x in said],
                            results_ok=bool(same and parted and said[0]["rows"] > 0),
                            checked="every rank holds the same merged frame; the bins were cut into equal-cost runs "
                                    "(sharding.split_counts): whole contigs decoded by one rank each, the contigs a cut falls "
                                    "into read as regions through the index")
        source.close_all()
        out = None
        if rank == 0:
            out = {"genome_frag_delfi_api_ranks": dict(
                ranks=world, fragments=rows_total, file_GB=round(os.path.getsize(pg) / 1e9, 3),
                file_and_side_files_write_s=round(t_write, 1), repetitions=reps, **best),
                "note": "one indexed file read by all ranks through frag.delfi; wall time of the slowest rank; PCIe and "
                        "the all-gather included; never the headline value"}
        grp.barrier()
        return out
    except Exception as exc:  # the headline line must still be printed
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        if rank == 0 and box[0]:
            shutil.rmtree(box[0][0], ignore_errors=True)


def end_to_end(torch, reps: int = 3, cpu=None):
    """File -> results on the host, through the product's own path (source.stream_source: host threads inflate,
    the GPU parses the rows, contigs become resident one after the other; then the kernels and the copy back).
    Four legs, files synthesised beforehand (not timed), `reps` repetitions each, the best one reported with its
    stage times:
      chr22_all_features  BASELINE configs 2/3: chr22 at 30x -> coverage + 1001-bin histogram + DELFI per 100 kb
                          window and WPS of every base, all results in host memory;
      delfi_4_contigs     config 4's shape on four contigs (19-22): DELFI short/long per 100 kb bin only;
      bam_60x_slice       config 5's input: a 60x paired-end BAM slice of 24 Mb (9.6 M records) -> read1 fragments ->
                          every feature and WPS of every base;
      genome_delfi_bins   config 4 itself: ONE whole-genome 30x frag.gz (309.6 M rows, 8 GB of text) -> DELFI
                          short / long / fragment counts of all 30 970 bins (two repetitions; FTK_BENCH_GENOME_E2E=0 skips);
      genome_all_features_wps  the same file -> every feature of every window and the WPS of every base on the host;
      genome_frag_delfi_api    the same file -> the product's frag.delfi() with its side files (frag_delfi_api_leg)."""
    import shutil
    import tempfile
    from finaletoolkit_amd import bgzf, source
    tmp = tempfile.mkdtemp(prefix="ftk_bench_")
    dev = torch.device("cuda", torch.cuda.current_device())
    res = {}
    try:
        threads = source.usable_cores()
        # The resident workload of the timed steps (60+ GB) was handed back to the driver a moment ago, and the first
        # large hipMalloc after such a release was seen to wait 0.3-0.6 s (FTK_WPS_TIMING: 0.57 s for the first leg's
        # 0.5 GB of scratch; never in a process that had not freed anything).  That wait is this script's, not the
        # path's, and it comes and goes (the driver cleans up behind the release): one second of rest and a 2 GB
        # allocation take it here, before any leg's clock starts; device_settle_s says how long the allocation waited.
        time.sleep(1.0)
        t_settle = time.perf_counter()
        pad = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
        pad.zero_()
        torch.cuda.synchronize()
        del pad
        torch.cuda.empty_cache()
        res["device_settle_s"] = round(time.perf_counter() - t_settle, 4)
        h2d = measure_h2d_gbs(torch, dev)
        LINK["d2h_GBps"] = round(measure_d2h_gbs(torch, dev), 1)
        LINK["host_write_GBps"] = round(measure_host_write_gbs(threads), 1)
        rates = inflate_alone_rates()

        def make_file(path, names):
            rows, truth = [], {}
            for c in names:
                size = synth.B37_SIZES[c]
                n = synth.n_fragments(size, 30.0)
                s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, size, n, synth.SEED_BASE + 100 + len(rows)))
                rows.append((c, s, e, q, st))
                ln = e - s
                truth[c] = dict(n=n, cov=int((q >= 30).sum()), delfi=int(((q >= 30) & (ln >= 100) & (ln <= 220)).sum()),
                                text_bytes=int(bgzf.row_lengths(c, s, e, q).sum()))
            t0 = time.perf_counter()
            bgzf.write_frag_gz(path, rows, level=1, with_index=False)
            t_write = time.perf_counter() - t0
            read_once(path)  # (untimed, see read_once)
            return truth, t_write

        def run(path, names, truth, all_features):
            runs = []
            for _ in range(reps):
                source.close_all()
                eng = source.get_engine()
                t0 = time.perf_counter()
                t_res = t_feat = t_wps = 0.0
                tb = t0
                ok = True
                n_win = 0
                src = None
                for src, c in source.stream_source(path, threads):
                    ta = time.perf_counter()
                    t_res += ta - tb
                    size = synth.B37_SIZES[c]
                    ws, we = synth.tiling_windows(size, WINDOW)
                    n_win += len(ws)
                    key = src.key(c)
                    if all_features:
                        # ONE launch per contig: feature blocks first, the WPS tiles behind them; everything to the host
                        r, w = eng.all_features_wps(key, ws, we, size, MAPQ, hist_bins=(0, HIST_BINS), delfi_q=MAPQ,
                                                    window_size=WPS_W, wps_min_length=WPS_MIN, wps_max_length=WPS_MAX,
                                                    wps_quality=MAPQ)
                        tf = tw = time.perf_counter()
                        ok = ok and int(r["coverage"].sum()) == truth[c]["cov"] and len(w) == size
                        ok = ok and int(r["short"].sum() + r["long"].sum()) == truth[c]["delfi"]
                        ok = ok and int(r["hist"].sum()) + int(r["overflow"].sum()) == truth[c]["cov"]
                        t_feat += tf - ta
                        t_wps += tw - tf
                        del w, r
                    else:
                        sh, lg, nf = eng.delfi_counts(key, ws, we, MAPQ, None, None, synth_gaps(size))
                        tf = time.perf_counter()
                        t_feat += tf - ta
                        ok = ok and bool(np.array_equal(sh + lg, nf)) and int(nf.sum()) <= truth[c]["delfi"]
                    tb = time.perf_counter()
                total = tb - t0
                cur = dict(total_s=round(total, 4), windows=n_win, windows_per_s=round(n_win / total, 1),
                           fragments_per_s_M=round(sum(t["n"] for t in truth.values()) / total / 1e6, 1),
                           waiting_for_resident_contigs_s=round(t_res, 4),
                           **({"features_wps_one_launch_and_copy_back_s": round(t_feat, 4)} if all_features
                              else {"feature_kernels_s": round(t_feat, 4)}),
                           decoder_producer_stage_ms=src.decode_stage_ms if src is not None else None, results_ok=bool(ok))
                runs.append(cur)
            leg = rep_summary(runs)
            leg_floor(leg, os.path.getsize(path), sum(t.get("text_bytes", 0) for t in truth.values()), "text", h2d, rates,
                      score_bases=sum(synth.B37_SIZES[c] for c in names) if all_features else 0)
            return leg

        p22 = os.path.join(tmp, "chr22.frag.gz")
        truth, t_write = make_file(p22, ["22"])
        res["chr22_all_features"] = dict(file_MB=round(os.path.getsize(p22) / 1e6, 1), fragments=truth["22"]["n"],
                                         file_write_s=round(t_write, 2), decoder_threads=threads,
                                         **run(p22, ["22"], truth, True))
        p4 = os.path.join(tmp, "c19_22.frag.gz")
        names = ["19", "20", "21", "22"]
        truth, t_write = make_file(p4, names)
        res["delfi_4_contigs"] = dict(file_MB=round(os.path.getsize(p4) / 1e6, 1), fragments=sum(t["n"] for t in truth.values()),
                                      file_write_s=round(t_write, 2), decoder_threads=threads,
                                      **run(p4, names, truth, False))
        # BASELINE config 5's input on one GPU: a 60x coordinate-sorted paired-end BAM slice (24 Mb, 9.6 M records)
        pb = os.path.join(tmp, "slice60x.bam")
        bsize = 24_000_000
        t0 = time.perf_counter()
        exp = synth.write_paired_bam(pb, "mid", bsize, 60.0, 31)
        t_write = time.perf_counter() - t0
        read_once(pb)  # (untimed, see read_once)
        ws, we = synth.tiling_windows(bsize, WINDOW)
        runs = []
        for _ in range(reps):
            source.close_all()
            eng = source.get_engine()
            t0 = time.perf_counter()
            for src, c in source.stream_source(pb, threads):
                t1 = time.perf_counter()
                key = src.key(c)
                r, w = eng.all_features_wps(key, ws, we, bsize, MAPQ, hist_bins=(0, HIST_BINS), delfi_q=MAPQ, window_size=WPS_W,
                                            wps_min_length=WPS_MIN, wps_max_length=WPS_MAX, wps_quality=MAPQ)  # ONE launch
                t2 = t3 = time.perf_counter()
            ok = (eng.info(key)[0] == exp["n"] and len(w) == bsize and
                  int(r["hist"].sum()) + int(r["overflow"].sum()) == int(r["coverage"].sum()) and int(r["coverage"].sum()) > 0)
            cur = dict(total_s=round(t3 - t0, 4), windows=len(ws), windows_per_s=round(len(ws) / (t3 - t0), 1),
                       fragments_per_s_M=round(exp["n"] / (t3 - t0) / 1e6, 1), waiting_for_resident_contigs_s=round(t1 - t0, 4),
                       features_wps_one_launch_and_copy_back_s=round(t3 - t1, 4),
                       decoder_producer_stage_ms=src.decode_stage_ms, results_ok=bool(ok))
            del w, r
            runs.append(cur)
        leg = rep_summary(runs)
        leg_floor(leg, exp["file_bytes"], 2 * exp["n"] * 115, "bam", h2d, rates, score_bases=bsize)  # (115-byte records: 50 bp reads)
        res["bam_60x_slice"] = dict(file_MB=round(exp["file_bytes"] / 1e6, 1), fragments=exp["n"], records=2 * exp["n"],
                                    file_write_s=round(t_write, 2), decoder_threads=threads, **leg)
        # BASELINE config 5 at real size: a > 4 GiB, three-contig 60x BAM (a chr1-sized contig between two small ones)
        if os.environ.get("FTK_BENCH_BIG_BAM", "1") != "0":
            res["bam_60x_chr1_scale"] = big_bam_leg(torch, tmp, threads, h2d, rates)
        # BASELINE config 5 as stated: the WHOLE genome as one 60x BAM, all features + WPS, one pass
        if os.environ.get("FTK_BENCH_GENOME_BAM", "1") != "0":
            rate = (res.get("bam_60x_chr1_scale") or {}).get("writer_records_per_s_M")
            res["bam_60x_genome"] = genome_bam_leg(torch, dev, threads, h2d, rates, rate * 1e6 if rate else None)
        # BASELINE config 4 itself, file to feature vector: ONE whole-genome 30x frag.gz -> DELFI bins of every contig
        if os.environ.get("FTK_BENCH_GENOME_E2E", "1") != "0":
            from finaletoolkit_amd import writers
            names = list(synth.B37_SIZES)
            n_win_total = sum(-(-synth.B37_SIZES[c] // WINDOW) for c in names)

            def write_genome(path, level):
                """ONE whole-genome 30x frag.gz at the given DEFLATE level (1: the files of rounds 1-3, fast to write;
                6: what bgzip / htslib write by default)."""
                t0 = time.perf_counter()
                truth, truth_all, rows_total, text_bytes = {}, {}, 0, 0
                for k, c in enumerate(names):
                    size = synth.B37_SIZES[c]
                    n = synth.n_fragments(size, 30.0)
                    s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, size, n, synth.SEED_BASE + k))
                    ln = e - s
                    truth[c] = int(((q >= MAPQ) & (ln >= 100) & (ln <= 220)).sum())
                    truth_all[c] = dict(n=n, cov=int((q >= 30).sum()), delfi=truth[c])
                    with writers.frag_rows(c, s, e, q, st) as text:
                        writers.bgzf_write(path, text, level, append=k > 0, write_eof=k == len(names) - 1)
                        text_bytes += text.n
                    rows_total += n
                    del s, e, q, st, ln
                open(path + ".tbi", "wb").close()  # (the reader streams the whole file; the index only has to exist)
                t_write = time.perf_counter() - t0
                read_once(path)  # (untimed, see read_once)
                return truth, truth_all, rows_total, text_bytes, t_write

            def genome_bins(path, truth, rows_total, text_bytes, t_write, rate_key):
                runs = []
                for _ in range(5):
                    source.close_all()
                    eng = source.get_engine()
                    t0 = time.perf_counter()
                    ok, seen, t_wait, tb, src = True, [], 0.0, t0, None
                    for src, c in source.stream_source(path, threads):
                        ta = time.perf_counter()
                        t_wait += ta - tb
                        size = synth.B37_SIZES[c]
                        ws, we = synth.tiling_windows(size, WINDOW)
                        sh, lg, nf = eng.delfi_counts(src.key(c), ws, we, MAPQ, None, None, synth_gaps(size))
                        ok = ok and bool(np.array_equal(sh + lg, nf)) and 0 < int(nf.sum()) <= truth[c]
                        seen.append(c)
                        tb = time.perf_counter()
                    total = tb - t0
                    runs.append(dict(total_s=round(total, 4), windows=n_win_total, windows_per_s=round(n_win_total / total, 1),
                                     fragments_per_s_M=round(rows_total / total / 1e6, 1),
                                     text_GB_per_s=round(text_bytes / total / 1e9, 2), waiting_for_resident_contigs_s=round(t_wait, 4),
                                     decoder_producer_stage_ms=src.decode_stage_ms if src is not None else None,
                                     results_ok=bool(ok and seen == names)))
                leg = rep_summary(runs)
                leg_floor(leg, os.path.getsize(path), text_bytes, rate_key, h2d, rates)
                return dict(file_GB=round(os.path.getsize(path) / 1e9, 2), text_GB=round(text_bytes / 1e9, 2), fragments=rows_total,
                            file_write_s=round(t_write, 1), decoder_threads=threads, **leg)

            pg = os.path.join(tmp, "genome.frag.gz")
            truth, truth_all, rows_total, text_bytes, t_write = write_genome(pg, 1)
            res["genome_delfi_bins"] = genome_bins(pg, truth, rows_total, text_bytes, t_write, "text")
            for c in names:
                truth_all[c]["text_bytes"] = 0
            truth_all[names[0]]["text_bytes"] = text_bytes  # (the file's text, for the floor of the next leg)
            # BASELINE config 5's shape on one GPU from the text file: every feature of every 100 kb window AND the WPS
            # of every base of the genome in host memory, contig by contig (24.8 GB of int64 scores cross PCIe)
            res["genome_all_features_wps"] = dict(file_GB=round(os.path.getsize(pg) / 1e9, 2), fragments=rows_total,
                                                  decoder_threads=threads,
                                                  scores_to_host_GB=round(8 * sum(synth.B37_SIZES.values()) / 1e9, 1),
                                                  **run(pg, names, truth_all, True))
            # The north-star's own figure, on the PRODUCT FUNCTION: the same file -> frag.delfi() with a bins file,
            # blacklist, gap annotation, a 2bit reference (per-bin GC counted on the device), the 100 kb -> 5 Mb
            # merge and the TSV written -- everything reference frag/_delfi.py:129-401 does around its per-bin loop.
            res["genome_frag_delfi_api"] = frag_delfi_api_leg(torch, tmp, pg, threads, n_win_total, rows_total,
                                                             res["genome_delfi_bins"], cpu)
            # The reference's OTHER commands through their product functions on the same file (tools/cmd_legs.py):
            # coverage(normalize=True) over the 30 970-row BED, multi_wps over 20 000 sites to .bw and to .bed.gz,
            # frag_length_intervals over the BED, genome-wide frag_length_bins - each with its stage split, a check
            # of one whole contig against the oracle (untimed) and the reference-shaped Python rate beside it.
            if os.environ.get("FTK_BENCH_CMD_LEGS", "1") != "0":
                try:
                    from tools import cmd_legs
                    res["commands"] = cmd_legs.measure(torch, tmp, pg, dict(synth.B37_SIZES), threads,
                                                       raw_floor=res["genome_delfi_bins"].get("floor"))
                except Exception as exc:  # noqa: BLE001 - the other legs must still be reported
                    res["commands"] = {"error": f"{type(exc).__name__}: {exc}"}
            os.remove(pg)
            os.remove(pg + ".tbi")
            # the same genome written at DEFLATE level 6 - what bgzip / htslib write by default, i.e. what a user's
            # frag.gz looks like (the level-1 file above is the one of rounds 1-3: 15 % larger, more and shorter matches)
            if os.environ.get("FTK_BENCH_LEVEL6", "1") != "0":
                pg6 = os.path.join(tmp, "genome_l6.frag.gz")
                truth6, _all6, rows6, text6, t_write6 = write_genome(pg6, 6)
                res["genome_delfi_bins_level6"] = dict(deflate_level=6, **genome_bins(pg6, truth6, rows6, text6, t_write6, "text_level6"))
                os.remove(pg6)
                os.remove(pg6 + ".tbi")
        res["note"] = ("every leg: first_s / median_s / best_s of its repetitions (total_s = best_s and the stage split are the "
                       "best repetition's; the first one of a process also pays thread-pool start, page-locked allocations and "
                       "the file's first read); floor = max(compressed bytes / measured H2D rate, the inflate kernel alone, and - legs that "
                       "return per-base scores - 2 B per base / measured D2H rate, 8 B per base / what the host cores write together); "
                       "PCIe transfers included; never the headline value")
        res["h2d_GBps_measured"] = round(h2d, 1)
        res["d2h_GBps_measured"] = LINK.get("d2h_GBps")
        res["host_write_GBps_measured"] = LINK.get("host_write_GBps")
        source.close_all()
    except Exception as exc:  # the headline line must still be printed
        res["error"] = f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def big_bam_leg(torch, tmp, threads, h2d, rates, reps: int = 3):
    """BASELINE config 5 at real size on one GPU (reference ``io/alignment.py:242-268`` over a whole-genome-scale BAM):
    a > 4 GiB coordinate-sorted 60x paired-end BAM of three contigs - a chr1-sized one (49.85 M pairs) between two small
    ones, 102.9 M records - written once (not timed), then streamed through the device inflate + device record parser
    (``source.stream_source``) ``reps`` times: every feature of every 100 kb window and the WPS of every base of every
    contig in host memory.  Checked (untimed, ``oracle/scale_check.py``): exact fragment counts, 24+ sampled windows per
    contig (coverage, histogram, DELFI) and 3 x 50 kb of WPS per contig against the C oracle in read1-fetch mode, the
    closed-form sum of every contig's WPS, and two region reads through the BAI whose file offsets lie BEHIND the
    4 GiB mark equal to the whole-contig answer."""
    from finaletoolkit_amd import source
    contigs = [("small_a", 3_000_000), ("big", synth.B37_SIZES["1"]), ("small_c", 5_000_000)]
    sizes = dict(contigs)
    path = os.path.join(tmp, "wg60x.bam")
    try:
        t0 = time.perf_counter()
        exp = synth.write_paired_bam_native(path, contigs, 60.0, 4242)
        t_write = time.perf_counter() - t0
        os.sync()  # (untimed: see genome_bam_leg)
        read_once(path)  # (untimed: the page cache's first read of a fresh file, see read_once)
        file_bytes = os.path.getsize(path)
        n_frag = sum(v["n"] for v in exp.values())
        n_win = sum(-(-n // WINDOW) for n in sizes.values())
        runs, feats, sums = [], {}, {}
        for rep in range(reps):
            source.close_all()
            eng = source.get_engine()
            t0 = time.perf_counter()
            t_wait = t_feat = t_wps = 0.0
            tb, seen, ok, src = t0, [], True, None
            for src, c in source.stream_source(path, threads):
                ta = time.perf_counter()
                t_wait += ta - tb
                ws, we = synth.tiling_windows(sizes[c], WINDOW)
                r, w = eng.all_features_wps(src.key(c), ws, we, sizes[c], MAPQ, hist_bins=(0, HIST_BINS), delfi_q=MAPQ,
                                            window_size=WPS_W, wps_min_length=WPS_MIN, wps_max_length=WPS_MAX,
                                            wps_quality=MAPQ)  # ONE launch: the read1 fetch rule on the fast kernels
                tf = tw = time.perf_counter()
                ok = ok and len(w) == sizes[c] and eng.info(src.key(c))[0] == exp[c]["n"]
                if rep == 0:
                    feats[c], sums[c] = r, int(w.sum())
                t_feat += tf - ta
                t_wps += tw - tf
                seen.append(c)
                del w
                tb = time.perf_counter()
            total = tb - t0
            runs.append(dict(total_s=round(total, 4), windows=n_win, windows_per_s=round(n_win / total, 1),
                             fragments_per_s_M=round(n_frag / total / 1e6, 1), file_GB_per_s=round(file_bytes / total / 1e9, 2),
                             waiting_for_resident_contigs_s=round(t_wait, 4),
                             features_wps_one_launch_and_copy_back_s=round(t_feat + t_wps, 4),
                             decoder_producer_stage_ms=src.decode_stage_ms,
                             results_ok=bool(ok and seen == [c for c, _ in contigs])))
        leg = rep_summary(runs)
        leg_floor(leg, file_bytes, 2 * n_frag * 125, "bam", h2d, rates, score_bases=sum(sizes.values()))  # (125-byte records: 50 bp reads, 10-byte names)
        # ---- the checks (untimed; the contigs of the last repetition are still resident) ----
        detail, ok = {}, leg["results_ok"] and file_bytes > (1 << 32)
        try:
            from oracle import scale_check as SC  # checker only
            eng = source.get_engine()
            for c, size in contigs:
                good, d = SC.check_contig(eng, src.key(c), size, exp[c], feats[c], n_sampled=24)
                d["wps_sum_ok"] = sums[c] == SC.wps_closed_form_sum(exp[c], size)
                detail[c] = d
                ok = ok and good and d["wps_sum_ok"]
            source.close_all()
            lazy = source.open_source(path)
            eng = source.get_engine()
            big = sizes["big"]
            a = big * 24 // 25 // WINDOW * WINDOW
            for c, r0, r1 in (("small_c", 2_000_000, 2_400_000), ("big", a, a + 400_000)):
                off = SC.region_file_offset(exp[c], r0)
                key = lazy.require_region(c, r0, r1)
                good, d = SC.check_region(eng, key, sizes[c], exp[c], r0, r1)
                d["file_offset"] = off
                d["behind_4GiB"] = off > (1 << 32)
                detail[f"region {c}:{r0}-{r1}"] = d
                ok = ok and good and d["behind_4GiB"] and c not in lazy.loaded
                lazy.release_region(key)
        except Exception as exc:  # noqa: BLE001
            ok, detail["error"] = False, f"{type(exc).__name__}: {exc}"
        leg["results_ok"] = bool(ok)
        return dict(file_GB=round(file_bytes / 1e9, 3), larger_than_4GiB=file_bytes > (1 << 32), contigs=len(contigs),
                    fragments=n_frag, records=2 * n_frag, file_write_s=round(t_write, 1),
                    writer_records_per_s_M=round(2 * n_frag / t_write / 1e6, 1), decoder_threads=threads, **leg, checked=detail)
    except Exception as exc:  # noqa: BLE001 - the other legs must still be reported
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        source.close_all()
        for q in (path, path + ".bai"):
            if os.path.exists(q):
                os.remove(q)


def genome_bam_leg(torch, dev, threads, h2d, rates, records_per_s, reps: int = 3):
    """BASELINE config 5 as stated: ONE whole-genome 60x coordinate-sorted paired-end BAM - all 24 b37 contigs, 6.2 x 10^8
    pairs / 1.24 x 10^9 records / ~71 GB at full scale; every contig shortened by the same factor when the box's scratch
    space or its writer's rate (two minutes at the rate the three-contig leg measured) cannot take that, ``scale`` says
    which - written once (not timed: ``synth.write_genome_bam``, records built, sorted and deflated by the host threads),
    then streamed ``reps`` times through the device inflate + device record parser (``source.stream_source``): for every
    contig every feature of every 100 kb window and the WPS of every base, ONE launch per contig (the read1 fetch rule of
    reference io/alignment.py:242-268 on the fast kernels), all results in host memory.  Checked (untimed,
    ``oracle/scale_check.py``): every contig's exact fragment count and the closed form of its WPS sum in every
    repetition; after the last one 24 sampled windows (coverage, histogram, DELFI) and 3 x 50 kb of WPS per contig
    against the C oracle in read1 mode on the still-resident contigs, and a region read through the BAI behind the offset
    at which the LAST contig begins."""
    import shutil
    import tempfile
    from finaletoolkit_amd import source
    d = None
    try:
        base = synth.big_scratch_dir(synth.genome_bam_bytes())
        d = tempfile.mkdtemp(prefix="ftk_wgbam_", dir=base)
        # How much of the genome the box can take: its scratch space, and two minutes of THIS writer on THIS box - measured
        # on a 2 % genome first (a second of writing; the three-contig leg's rate comes from another writer path and
        # undersold this one threefold, which kept the file at 0.66 of the genome on boxes that hold all of it).
        if not os.environ.get("FTK_WG_BAM_SCALE"):
            probe = os.path.join(d, "probe.bam")
            t0 = time.perf_counter()
            _, pinfo = synth.write_genome_bam(probe, 0.02, 60.0, torch, dev)
            records_per_s = 2 * sum(v["n"] for v in pinfo.values()) / (time.perf_counter() - t0)
            for q in (probe, probe + ".bai"):
                if os.path.exists(q):
                    os.remove(q)
        scale = synth.genome_bam_scale(d, records_per_s=records_per_s, write_budget_s=120.0)
        path = os.path.join(d, "genome60x.bam")
        t0 = time.perf_counter()
        contigs, info = synth.write_genome_bam(path, scale, 60.0, torch, dev)
        t_write = time.perf_counter() - t0
        # (untimed) the file's dirty pages go to the disk NOW: the first pass must not share the memory bus and the page
        # cache's locks with the write-back of the tens of GB it is about to read - first_s is a reader's first call,
        # not a writer's aftermath
        t0 = time.perf_counter()
        os.sync()
        t_sync = time.perf_counter() - t0
        t_preread = read_once(path)
        sizes = dict(contigs)
        names = [c for c, _ in contigs]
        file_bytes = os.path.getsize(path)
        n_frag = sum(v["n"] for v in info.values())
        n_win = sum(-(-n // WINDOW) for n in sizes.values())
        from oracle import scale_check as SC  # checker only
        runs, sums, src = [], {}, None
        for rep in range(reps):
            source.close_all()
            eng = source.get_engine()
            t0 = time.perf_counter()
            t_wait = t_launch = 0.0
            tb, seen, ok = t0, [], True
            for src, c in source.stream_source(path, threads):
                ta = time.perf_counter()
                t_wait += ta - tb
                ws, we = synth.tiling_windows(sizes[c], WINDOW)
                r, w = eng.all_features_wps(src.key(c), ws, we, sizes[c], MAPQ, hist_bins=(0, HIST_BINS), delfi_q=MAPQ,
                                            window_size=WPS_W, wps_min_length=WPS_MIN, wps_max_length=WPS_MAX, wps_quality=MAPQ)
                tl = time.perf_counter()
                t_launch += tl - ta
                ok = ok and len(w) == sizes[c] and eng.info(src.key(c))[0] == info[c]["n"]
                ok = ok and int(r["hist"].sum()) + int(r["overflow"].sum()) == int(r["coverage"].sum()) > 0
                if rep == reps - 1:
                    sums[c] = (int(w.sum()), r)
                seen.append(c)
                del w
                tb = time.perf_counter()  # (the sums above are the bench's own bookkeeping: outside the clock's stages, inside its total)
            total = tb - t0
            runs.append(dict(total_s=round(total, 4), windows=n_win, windows_per_s=round(n_win / total, 1),
                             fragments_per_s_M=round(n_frag / total / 1e6, 1), file_GB_per_s=round(file_bytes / total / 1e9, 2),
                             waiting_for_resident_contigs_s=round(t_wait, 4),
                             features_wps_one_launch_and_copy_back_s=round(t_launch, 4),
                             decoder_producer_stage_ms=src.decode_stage_ms, results_ok=bool(ok and seen == names)))
        leg = rep_summary(runs)
        leg_floor(leg, file_bytes, 2 * n_frag * 125, "bam", h2d, rates, score_bases=sum(sizes.values()))
        detail, ok = {}, leg["results_ok"]
        try:
            eng = source.get_engine()
            n_checked = wps_checked = 0
            for k, (c, size) in enumerate(contigs):
                exp = synth.genome_bam_expected(k, size, 60.0, torch, dev)
                good, dd = SC.check_contig(eng, src.key(c), size, exp, sums[c][1], n_sampled=28)
                good = good and sums[c][0] == SC.wps_closed_form_sum(exp, size)
                n_checked += dd["windows_checked"]
                wps_checked += dd["wps_bases_checked"]
                if not good:
                    detail[c] = dd
                ok = ok and good
                if c == names[-1]:
                    last_exp = exp
            detail["contigs_checked"] = len(contigs)
            detail["windows_checked"] = n_checked
            detail["wps_bases_checked"] = wps_checked
            detail["every_contig_wps_sum_equals_closed_form"] = bool(ok)
            source.close_all()
            lazy = source.open_source(path)
            eng = source.get_engine()
            c, size = contigs[-1]
            a = size // 2 // WINDOW * WINDOW
            off = SC.region_file_offset(dict(linear=info[c]["linear"]), a)
            key = lazy.require_region(c, a, a + 4 * WINDOW)
            good, dd = SC.check_region(eng, key, size, last_exp, a, a + 4 * WINDOW)
            dd["file_offset"] = off
            dd["behind_the_last_contigs_offset"] = bool(off >= info[c]["first_off"])
            dd["behind_4GiB"] = bool(off > (1 << 32))
            detail[f"region {c}:{a}-{a + 4 * WINDOW}"] = dd
            ok = ok and good and dd["behind_the_last_contigs_offset"] and c not in lazy.loaded
        except Exception as exc:  # noqa: BLE001
            ok, detail["error"] = False, f"{type(exc).__name__}: {exc}"
        leg["results_ok"] = bool(ok)
        return dict(scale=scale, full_genome=bool(scale >= 1.0), contigs=len(contigs), file_GB=round(file_bytes / 1e9, 2),
                    scratch=base, fragments=n_frag, records=2 * n_frag, file_write_s=round(t_write, 1), file_sync_s=round(t_sync, 1), file_read_once_s=round(t_preread, 2),
                    writer_GB_per_s=round(file_bytes / t_write / 1e9, 2), decoder_threads=threads, **leg, checked=detail)
    except Exception as exc:  # noqa: BLE001 - the other legs must still be reported
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        source.close_all()
        if d:
            shutil.rmtree(d, ignore_errors=True)


def frag_delfi_api_leg(torch, tmp, genome_file, threads, n_win_total, rows_total, raw_leg, cpu):
    """`frag.delfi()` on the whole-genome 30x file (30 970 x 100 kb bins merged to 5 Mb): wall time with the stage split
    of frag/_delfi.py's LAST_STAGE_S, the ratio to the raw engine leg, the x-factor against the reference-shaped
    single-thread Python rate of the same bins (cpu_baseline), and a check of one whole contig of the unmerged frame
    against the C oracle (checker only)."""
    import pandas
    from finaletoolkit_amd import frag, source
    from finaletoolkit_amd.frag import _delfi as FD
    sizes = dict(synth.B37_SIZES)
    t0 = time.perf_counter()
    cs, bins, bl, gapbed = synth.write_genome_delfi_inputs(tmp, sizes, WINDOW)
    ref2bit = os.path.join(tmp, "genome.2bit")
    synth.write_random_2bit(ref2bit, sizes)
    t_inputs = time.perf_counter() - t0
    out_tsv = os.path.join(tmp, "delfi_5mb.tsv")
    runs, df = [], None
    import warnings
    for _ in range(5):
        source.close_all()
        ta = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            df = frag.delfi(genome_file, cs, bins, ref2bit, blacklist_file=bl, gap_file=gapbed, output_file=out_tsv,
                            no_gc_correct=True, merge_bins=True, workers=threads)
        total = time.perf_counter() - ta
        runs.append(dict(total_s=round(total, 4), stages_s=dict(FD.LAST_STAGE_S)))
    best = rep_summary(runs)
    best.pop("results_ok", None)
    # checker (untimed): the unmerged frame of one whole contig against the C oracle, and the merged file's shape
    ok = df.shape[0] > 400 and os.path.getsize(out_tsv) > 10_000 and list(df.columns)[:4] == ["contig", "start", "stop", "arm"]
    try:
        from oracle import oracle as O
        c, ci = "21", list(sizes).index("21")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            un = frag.delfi(genome_file, cs, bins, ref2bit, blacklist_file=bl, gap_file=gapbed, no_gc_correct=True,
                            merge_bins=False, remove_nocov=False, workers=threads)
        rows = un.loc[un["contig"] == c]
        dev = torch.device("cuda", torch.cuda.current_device())
        s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, sizes[c], synth.n_fragments(sizes[c], 30.0),
                                                                 synth.SEED_BASE + ci))
        b = pandas.read_csv(bl, sep="\t", names=["c", "s", "e"], dtype={"c": str})
        b = b.loc[b["c"] == c].sort_values(["s", "e"])
        sh, lg, nf = O.c_delfi_counts(O.Frags(s, e, q, st), rows["start"].to_numpy(), rows["stop"].to_numpy(), MAPQ,
                                      b["s"].to_numpy(), b["e"].to_numpy(), synth_gaps(sizes[c]))
        ok = ok and len(rows) > 300 and bool(np.array_equal(rows["short"].to_numpy().astype(np.int64), sh)
                                             and np.array_equal(rows["long"].to_numpy().astype(np.int64), lg)
                                             and np.array_equal(rows["num_frags"].to_numpy(), nf))
        checked = f"contig {c}: {len(rows)} unmerged rows equal to the C oracle"
    except Exception as exc:  # noqa: BLE001
        ok, checked = False, f"{type(exc).__name__}: {exc}"
    leg = dict(bins_100kb=n_win_total, merged_rows=int(df.shape[0]), fragments=rows_total, decoder_threads=threads,
               side_files_write_s=round(t_inputs, 2), **best,
               windows_per_s=round(n_win_total / best["total_s"], 1),
               vs_raw_engine_leg=round(best["total_s"] / raw_leg["total_s"], 3), results_ok=bool(ok), checked=checked)
    if raw_leg.get("floor"):  # the same file: the same floor
        f = dict(raw_leg["floor"])
        f["frac_of_floor_best"] = round(f["floor_s"] / leg["best_s"], 3)
        f["frac_of_floor_median"] = round(f["floor_s"] / leg["median_s"], 3)
        leg["floor"] = f
    py = (cpu or {}).get("reference_shaped_python_delfi")
    if py:
        # the target of BASELINE.json's north_star: >= 50x the reference's single-thread rate on these bins
        leg["reference_shaped_python_windows_per_s"] = py["value"]
        leg["x_vs_reference_shaped_python"] = round(n_win_total / best["total_s"] / py["value"], 1)
    return leg


def cpu_baseline(torch, eng, per, c, budget_s, checks):
    """C oracle (kind "port", 1 core) on a bounded sample of the same workload (unit ``c``), checked against the GPU
    results for the same windows; the same restatement on every host core; and the reference-shaped single-thread
    Python restatement on BASELINE.md section 3's minimum sample (2 000 windows of counters, 200 WPS tiles)."""
    from oracle import oracle as O
    p = per[c]
    size, a0 = p["size"], p["a"]  # a partial unit carries every fragment its windows / bases can see
    s, e, q, st = [t.cpu().numpy() for t in p["keep"]]
    fr = O.Frags(s, e, q, st)
    n_s = min(p["nw"], 600)
    ws, we = p["ws"][:n_s], p["we"][:n_s]
    t0 = time.perf_counter()
    cov = O.c_window_counts(fr, ws, we, mapq_min=MAPQ)
    hist, over = O.c_fraglen_hist(fr, ws, we, 0, HIST_BINS, mapq_min=MAPQ)
    sh, lg, nf = O.c_delfi_counts(fr, ws, we, MAPQ, p["bl"][0], p["bl"][1], p["gaps"])
    t_count = time.perf_counter() - t0
    # WPS in 5 kb tiles like multi_wps; stop when the budget is used, extrapolate per window
    done = 0
    t1 = time.perf_counter()
    ok_wps = True
    wps_gpu = None
    while done < n_s and (time.perf_counter() - t1) < budget_s:
        a, b = int(ws[done]), int(we[done])
        ref = np.concatenate([O.c_wps(fr, x, min(x + 5000, b), size, WPS_W, WPS_MIN, WPS_MAX, MAPQ)
                              for x in range(a, b, 5000)])
        wps_gpu = p["wps"][a - a0:b - a0].cpu().numpy()
        ok_wps = ok_wps and bool(np.array_equal(ref, wps_gpu))
        done += 1
    t_wps = time.perf_counter() - t1
    per_window = t_count / n_s + t_wps / max(done, 1)
    checks["sample_cov"] = bool(np.array_equal(cov, p["cov"][:n_s].cpu().numpy()))
    checks["sample_hist"] = bool(np.array_equal(hist.astype(np.int64), p["hist"][:n_s].cpu().numpy().astype(np.int64)))
    checks["sample_delfi"] = bool(np.array_equal(sh, p["short"][:n_s].cpu().numpy())
                                  and np.array_equal(lg, p["long"][:n_s].cpu().numpy()))
    checks["sample_wps"] = ok_wps
    # the same C restatement on every host core: one 100 kb window per task, pthreads inside the oracle library
    from finaletoolkit_amd.source import usable_cores
    n_cores = usable_cores()
    n_all = int(min(p["nw"], max(64, 8 * n_cores)))
    t_all = O.c_all_cores(fr, p["ws"][:n_all], p["we"][:n_all], HIST_BINS, MAPQ, p["bl"][0], p["bl"][1], p["gaps"], size,
                          WPS_W, WPS_MIN, WPS_MAX, n_cores)
    # reference-shaped single-thread baseline: the pure-Python restatement that
    # follows the reference loop for loop.  Every window / WPS tile gets the rows its index query would return
    # (the contig's fragments overlapping it, at full depth); as many windows and tiles as fit in ~6 s each.
    s64, e64 = s.astype(np.int64), e.astype(np.int64)

    def fetched(a, b):  # rows of a tabix query [a, b): start < b and end > a
        lo = int(np.searchsorted(s64, a - 1000, side="left"))  # fragments are at most 1000 bp long
        hi = int(np.searchsorted(s64, b, side="left"))
        k = np.flatnonzero(e64[lo:hi] > a) + lo
        return list(zip(s[k].tolist(), e[k].tolist(), q[k].tolist(), st[k].tolist()))

    bl_rows = list(zip(p["bl"][0].tolist(), p["bl"][1].tolist()))
    # >= 2 000 windows for the counters, >= 200 x 5 kb tiles for WPS (a wall cap only guards a
    # pathologically slow host: the sample is then smaller and says so)
    PY_WINDOWS, PY_TILES, PY_CAP_S = 2000, 200, 240.0
    ws_py, we_py = p["ws"], p["we"]
    n_py_want = min(p["nw"], PY_WINDOWS)
    t2 = time.perf_counter()
    n_py = 0
    t_py_delfi_sum = 0.0
    while n_py < n_py_want and time.perf_counter() - t2 < PY_CAP_S:
        a, b = int(ws_py[n_py]), int(we_py[n_py])
        tf = time.perf_counter()
        rows = fetched(a, b)
        tf = time.perf_counter() - tf
        O.py_single_coverage(rows, a, b, None, None, "midpoint", MAPQ)
        O.py_distribution(rows, a, b, None, None, "midpoint", MAPQ)
        # DELFI alone (frag/_delfi.py:404-511 per 100 kb bin: fetch + per-fragment Python loop) is also what the
        # whole-genome frag.delfi() leg of end_to_end is set against: timed by itself inside the same pass
        td = time.perf_counter()
        O.py_delfi_single_window(rows, a, b, MAPQ, bl_rows, p["gaps"])
        t_py_delfi_sum += time.perf_counter() - td + tf
        n_py += 1
    t_py_count = (time.perf_counter() - t2) / max(n_py, 1)
    n_pyd = n_py
    t_py_delfi = t_py_delfi_sum / max(n_py, 1)
    t3 = time.perf_counter()
    n_tiles_py = 0
    x0 = int(ws[min(1, n_s - 1)])
    while n_tiles_py < PY_TILES and time.perf_counter() - t3 < PY_CAP_S:
        a = x0 + 5000 * n_tiles_py
        if a + 5000 > size:
            break
        O.py_wps(fetched(max(a - WPS_MAX, 0), a + 5000 + WPS_MAX), a, a + 5000, size, WPS_W, WPS_MIN, WPS_MAX, MAPQ)
        n_tiles_py += 1
    t_py_wps = (time.perf_counter() - t3) / max(n_tiles_py, 1) * (WINDOW / 5000)
    py_sample_s = time.perf_counter() - t2
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(1.0 / per_window, 3), "unit": "windows/s", "cores": 1, "kind": "port", "host_cpu": cpu_model,
            "sample": f"C oracle (oracle/ftk_oracle.c, gcc -O2): coverage+hist+DELFI on {n_s} and WPS (5 kb tiles) on "
                      f"{done} x 100 kb windows of contig {c}; extrapolated per window",
            "all_cores": {"value": round(n_all / t_all, 2), "unit": "windows/s", "cores": n_cores, "kind": "port",
                          "sample": f"the same C restatement, one 100 kb window per task on {n_cores} pthreads (usable cores of "
                                    f"{os.cpu_count()} logical CPUs), "
                                    f"{n_all} windows of contig {c}"},
            "reference_shaped_python_delfi": {
                "value": round(1.0 / t_py_delfi, 3), "unit": "windows/s", "cores": 1,
                "sample": f"oracle/oracle.py py_delfi_single_window on the rows an index query returns at full depth, "
                          f"{n_pyd} x 100 kb bins of contig {c} (a 5 Mb bin = 50 of them)"},
            "reference_shaped_python": {
                "value": round(1.0 / (t_py_count + t_py_wps), 4), "unit": "windows/s", "cores": 1,
                "sample": f"oracle/oracle.py py_* (per-window fetch + per-fragment Python predicate, numpy "
                          f"_single_nt_wps) on the rows an index query returns at full depth: counters of {n_py} windows, "
                          f"{n_tiles_py} x 5 kb WPS tiles of unit {c} ({py_sample_s:.0f} s of CPU work; BASELINE.md section 3 "
                          f"asks for >= 2000 / >= 200); per 100 kb window = counters + 20 tiles, extrapolated linearly"}}


if __name__ == "__main__":
    main()
