"""
The exchange between the ranks of one node, behind one small interface.

The reference fans its windows out over ``multiprocessing.Pool(workers)`` and gets one list back
(frag/_delfi.py:289-300, frag/_coverage.py:212-248); here the fan-out is one process per GPU and what comes back is an
all-gather of fixed-width int64 rows plus, for ``coverage(normalize=True)``, one int64 all-reduce.  Two transports:

``RcclGroup``   the library's own communicator (``ftk_comm_*`` in include/ftk.h: RCCL over xGMI, resolved with dlopen,
                no torch in the process).  The default for N ranks on N GPUs.  Ranks meet through a small file
                (``FTK_COMM_ID_FILE``, else ``<tmp>/ftk_comm_<launcher pid>_<MASTER_PORT>.id``).
``TorchGroup``  ``torch.distributed`` (gloo): CPU tests and several ranks sharing ONE GPU, which RCCL refuses.  Also what
                a caller gets who initialised ``torch.distributed`` themselves before calling into the package.

Small Python objects (error strings, payload sizes) travel as pickled bytes in int64 words through the same
all-gather: the collectives of the C ABI are all the transport a sharded command needs.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
import tempfile
from typing import List, Optional

import numpy as np

_GROUP = None  # the process's group once joined


class Group:
    """One process: every exchange is the identity."""
    backend = "none"
    rank, world = 0, 1

    def all_gather_i64(self, send: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(send, dtype=np.int64).reshape(1, -1)

    def all_reduce_sum_i64(self, values: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(values, dtype=np.int64).copy()

    def all_gather_object(self, obj) -> list:
        return [obj]

    def broadcast_object(self, obj, src: int = 0):
        return obj

    def barrier(self) -> None:
        pass

    def send_bytes(self, data, dst: int) -> None:
        raise RuntimeError("a single process has nobody to send to")

    def recv_bytes(self, n: int, src: int) -> np.ndarray:
        raise RuntimeError("a single process has nobody to receive from")

    def close(self) -> None:
        pass

    # -- built on the two primitives ------------------------------------------------------------------------------
    def _gather_blobs(self, blob: bytes) -> List[bytes]:
        sizes = self.all_gather_i64(np.array([len(blob)], np.int64)).reshape(-1)
        words = max(int(-(-int(sizes.max()) // 8)), 1)
        buf = np.zeros(words * 8, np.uint8)
        buf[:len(blob)] = np.frombuffer(blob, np.uint8)
        got = self.all_gather_i64(buf.view(np.int64))
        return [got[r].view(np.uint8)[:int(sizes[r])].tobytes() for r in range(self.world)]


class RcclGroup(Group):
    """``ftk_comm_*``: RCCL, one rank per GPU.  Owns a context of its own on the rank's device (an engine that is closed
    and reopened between files must not take the communicator with it)."""
    backend = "rccl"

    def __init__(self, rank: int, world: int, device: int, id_hex_or_path: Optional[str]):
        from . import _lib as L
        from .engine import Engine
        self.L = L
        self.lib = L.load()
        self.eng = Engine(device)
        self.device = device
        h = C.c_void_p()
        ident = None if id_hex_or_path is None else str(id_hex_or_path).encode()
        if job_nonce():  # (the library reads the launch's nonce from the environment: ftk_comm_create)
            os.environ["FTK_COMM_NONCE"] = job_nonce()
        rc = self.lib.ftk_comm_create(self.eng.ctx, int(rank), int(world), ident, C.byref(h))
        if rc != L.FTK_OK:
            msg = self.lib.ftk_last_error(self.eng.ctx).decode()
            self.eng.close()
            raise L.FtkError(rc, msg)
        self.h = h
        r, w = C.c_int(), C.c_int()
        self.lib.ftk_comm_size(self.h, C.byref(r), C.byref(w))
        self.rank, self.world = int(r.value), int(w.value)

    def _check(self, rc):
        if rc != self.L.FTK_OK:
            raise self.L.FtkError(rc, self.lib.ftk_last_error(self.eng.ctx).decode())

    def set_stream(self, hip_stream) -> None:
        """Order the collectives behind (and ``join`` them into) this HIP stream instead of the group's own."""
        self.eng.set_stream(hip_stream)

    def all_gather_i64(self, send: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(send, dtype=np.int64).reshape(-1)
        out = np.zeros((self.world, len(s)), np.int64)
        if len(s):
            self._check(self.lib.ftk_allgather_i64(self.h, self.L.ptr(s), len(s), self.L.ptr(out)))
        return out

    def all_gather_i64_device(self, send_ptr, n: int, recv_ptr) -> None:
        """Device buffers (addresses or tensors), stream-ordered; ``join()`` before the results are read on the stream."""
        self._check(self.lib.ftk_allgather_i64(self.h, self.L.ptr(send_ptr), int(n), self.L.ptr(recv_ptr)))

    def join(self) -> None:
        self._check(self.lib.ftk_comm_join(self.h))

    def all_reduce_sum_i64(self, values: np.ndarray) -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.int64).reshape(-1).copy()
        if len(v):
            self._check(self.lib.ftk_allreduce_sum_i64(self.h, self.L.ptr(v), len(v)))
        return v

    def all_gather_object(self, obj) -> list:
        return [pickle.loads(b) for b in self._gather_blobs(pickle.dumps(obj))]

    def broadcast_object(self, obj, src: int = 0):
        return self.all_gather_object(obj if self.rank == src else None)[src]

    def barrier(self) -> None:
        self.all_reduce_sum_i64(np.ones(1, np.int64))

    def send_bytes(self, data, dst: int) -> None:
        a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
        if len(a):
            self._check(self.lib.ftk_comm_send(self.h, int(dst), self.L.ptr(np.ascontiguousarray(a)), len(a)))

    def recv_bytes(self, n: int, src: int) -> np.ndarray:
        out = np.empty(int(n), np.uint8)
        if n:
            self._check(self.lib.ftk_comm_recv(self.h, int(src), self.L.ptr(out), int(n)))
        return out

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.ftk_comm_destroy(self.h)
            self.h = None
            self.eng.close()


class TorchGroup(Group):
    """``torch.distributed`` (gloo in the CPU tests and when ranks share a GPU; whatever the caller initialised)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = "torch/" + dist.get_backend(group)

    def _device(self):
        import torch
        if self.dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return None

    def all_gather_i64(self, send: np.ndarray) -> np.ndarray:
        import torch
        s = torch.from_numpy(np.ascontiguousarray(send, dtype=np.int64).reshape(-1).copy())
        if len(s) == 0:
            return np.zeros((self.world, 0), np.int64)
        dev = self._device()
        if dev is not None:
            s = s.to(dev)
        recv = [torch.zeros_like(s) for _ in range(self.world)]
        self.dist.all_gather(recv, s, group=self.group)
        return np.stack([t.cpu().numpy() for t in recv])

    def all_reduce_sum_i64(self, values: np.ndarray) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.int64).reshape(-1).copy())
        dev = self._device()
        if dev is not None:
            t = t.to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def all_gather_object(self, obj) -> list:
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def broadcast_object(self, obj, src: int = 0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def barrier(self) -> None:
        self.dist.barrier(group=self.group)

    def send_bytes(self, data, dst: int) -> None:
        import torch
        a = np.frombuffer(data, np.uint8).copy() if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
        dev = self._device()
        for o in range(0, len(a), 1 << 30):
            t = torch.from_numpy(a[o:o + (1 << 30)])
            self.dist.send(t.to(dev) if dev is not None else t, dst=dst, group=self.group)

    def recv_bytes(self, n: int, src: int) -> np.ndarray:
        import torch
        out = np.empty(int(n), np.uint8)
        dev = self._device()
        for o in range(0, int(n), 1 << 30):
            m = min(1 << 30, int(n) - o)
            t = torch.empty(m, dtype=torch.uint8, device=dev if dev is not None else "cpu")
            self.dist.recv(t, src=src, group=self.group)
            out[o:o + m] = t.cpu().numpy()
        return out

    def close(self) -> None:
        if self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()


def job_nonce() -> str:
    """What tells this launch's rendezvous file from any other's: ``FTK_COMM_NONCE`` (``sharding.launch_ranks`` makes
    one per launch), else the launcher's run id (``TORCHELASTIC_RUN_ID`` when it is not torchrun's default ``none``),
    else nothing - the library then relies on the file's age alone (``ftk_comm_create``)."""
    n = os.environ.get("FTK_COMM_NONCE")
    if n:
        return n
    run = os.environ.get("TORCHELASTIC_RUN_ID", "")
    return run if run and run != "none" else ""


def id_file() -> str:
    """Where the ranks of this job meet: ``FTK_COMM_ID_FILE`` (``sharding.launch_ranks`` exports one inside a fresh
    directory of its own), else a name all ranks of one launch derive alike from what the LAUNCHER gave them - its run
    id when it set one, MASTER_ADDR / MASTER_PORT - and, only without a run id, the ranks' common parent pid (a launcher
    that puts a wrapper process around every rank must therefore set ``FTK_COMM_ID_FILE`` or a run id)."""
    p = os.environ.get("FTK_COMM_ID_FILE")
    if p:
        return p
    run = job_nonce()
    tag = run if run else str(os.getppid())
    tag = "".join(ch if ch.isalnum() else "_" for ch in tag)[:64]
    port = os.environ.get("MASTER_PORT", "0")
    return os.path.join(tempfile.gettempdir(), f"ftk_comm_{os.getuid()}_{tag}_{port}.id")


def current(group=None) -> Group:
    """The group this process exchanges in: the one ``join`` made, a ``torch.distributed`` group the caller initialised
    (only looked for when ``torch.distributed`` is imported already - a single-process call never imports torch), else
    the one-process identity."""
    import sys
    if group is not None and not isinstance(group, Group):
        return TorchGroup(group)
    if isinstance(group, Group):
        return group
    if _GROUP is not None:
        return _GROUP
    if "torch.distributed" in sys.modules:
        dist = sys.modules["torch.distributed"]
        if dist.is_available() and dist.is_initialized():
            return TorchGroup()
    return Group()


def join(backend: Optional[str] = None) -> Group:
    """Join the job's group from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; ``torchrun``,
    ``python -m finaletoolkit_amd.cli --gpus N``, ``sharding.launch_ranks``).  ``FTK_DIST_BACKEND`` / ``backend``:
    ``rccl`` (default: the library's communicator, GPU ``FTK_DEVICE`` else ``LOCAL_RANK``), ``gloo`` or ``nccl``
    (``torch.distributed``).  One process: nothing is started."""
    global _GROUP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return Group()
    if _GROUP is not None:
        return _GROUP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL between processes: the host driver only has dmabuf IPC)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or os.environ.get("FTK_DIST_BACKEND", "rccl")
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("FTK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if backend == "rccl":
        from . import _lib as L
        lib = L.load()
        n = C.c_int(0)
        lib.ftk_device_count(C.byref(n))
        if n.value <= local:
            raise RuntimeError(f"rank {rank}: GPU {local} is not visible ({n.value} devices); one rank per GPU, no fallback")
        _GROUP = RcclGroup(rank, world, local, id_file())
        return _GROUP
    import datetime
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        limit = datetime.timedelta(seconds=float(os.environ.get("FTK_DIST_TIMEOUT_S", "1800")))
        if backend == "nccl":
            from . import _lib
            _lib._hardware_queues()  # torch starts the HIP runtime below; the decoder's queue count must be set before
            if torch.cuda.device_count() <= local:
                raise RuntimeError(f"rank {rank}: GPU {local} is not visible ({torch.cuda.device_count()} devices); "
                                   f"one rank per GPU, no fallback")
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=limit)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)
    _GROUP = TorchGroup()
    return _GROUP


def leave() -> None:
    """Leave the group ``join`` made (end of a multi-rank command)."""
    global _GROUP
    g, _GROUP = _GROUP, None
    if g is not None:
        g.barrier()
        g.close()
