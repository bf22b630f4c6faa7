"""
The exchange layer (finaletoolkit_amd/comm.py; C ABI: ftk_comm_* in include/ftk.h).

CPU: the one-process identity, the rendezvous file name, and ``TorchGroup`` over gloo with two ranks (the transport
of the CPU tests and of ranks sharing a GPU) - objects, rows, payload bytes.
GPU: the library's RCCL communicator through ctypes with one rank (the build boxes have one GPU: RCCL refuses two
ranks on one device) - host and device buffers, the rendezvous through a file, the object gather built on top.
The reference's counterpart is the result list its ``Pool`` hands back: frag/_delfi.py:289-300, frag/_coverage.py:215-227.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_process_group_is_the_identity_and_needs_no_torch():
    code = """
import sys, numpy as np
from finaletoolkit_amd import comm, sharding
g = comm.current()
assert (g.rank, g.world, g.backend) == (0, 1, "none")
assert np.array_equal(g.all_gather_i64(np.arange(5)), np.arange(5).reshape(1, 5))
assert g.all_gather_object({"a": 1}) == [{"a": 1}] and g.broadcast_object(7) == 7
assert int(g.all_reduce_sum_i64(np.array([41]))[0]) == 41
assert sharding.rank_world() == (0, 1) and sharding.allreduce_sum(9) == 9
assert sharding.init_from_env() == (0, 1)
sharding.agree(None)
assert "torch" not in sys.modules
print("ok")
"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_rendezvous_file_name(monkeypatch, tmp_path):
    from finaletoolkit_amd import comm
    for k in ("FTK_COMM_ID_FILE", "FTK_COMM_NONCE", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_PORT", "4711")
    assert comm.id_file().endswith(f"ftk_comm_{os.getuid()}_{os.getppid()}_4711.id") and comm.job_nonce() == ""
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")  # torchrun's default says nothing about the launch
    assert comm.id_file().endswith(f"ftk_comm_{os.getuid()}_{os.getppid()}_4711.id")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job 17/a")  # a launcher-given run id names the file, not the parent pid
    assert comm.id_file().endswith(f"ftk_comm_{os.getuid()}_job_17_a_4711.id") and comm.job_nonce() == "job 17/a"
    monkeypatch.setenv("FTK_COMM_NONCE", "abc")
    assert comm.job_nonce() == "abc"
    monkeypatch.setenv("FTK_COMM_ID_FILE", str(tmp_path / "x.id"))
    assert comm.id_file() == str(tmp_path / "x.id")


def test_launch_ranks_gives_every_launch_its_own_meeting_place(tmp_path):
    """sharding.launch_ranks: a fresh directory only this user can enter, the file name and a nonce in the ranks'
    environment; the directory is gone when the launch is over."""
    from finaletoolkit_amd import sharding
    out = tmp_path / "env"
    code = ("import os, stat; p = os.environ['FTK_COMM_ID_FILE']; d = os.path.dirname(p); "
            f"open(r'{out}' + os.environ['RANK'], 'w').write('|'.join([p, os.environ['FTK_COMM_NONCE'], oct(stat.S_IMODE(os.stat(d).st_mode))]))")
    env = {k: os.environ.pop(k) for k in ("FTK_COMM_ID_FILE", "FTK_COMM_NONCE") if k in os.environ}
    try:
        assert sharding.launch_ranks([sys.executable, "-c", code], 2, share_gpu=True) == 0
        a = open(str(out) + "0").read().split("|")
        b = open(str(out) + "1").read().split("|")
        assert sharding.launch_ranks([sys.executable, "-c", code], 2, share_gpu=True) == 0
        c = open(str(out) + "0").read().split("|")
    finally:
        os.environ.update(env)
    assert a == b and a[2] == "0o700" and len(a[1]) == 32 and not os.path.exists(os.path.dirname(a[0]))
    assert c[0] != a[0] and c[1] != a[1]


_WORKER = """
import os, sys, numpy as np
sys.path.insert(0, os.environ["FTK_ROOT"])
from finaletoolkit_amd import comm, sharding
g = comm.join("gloo")
r, w = g.rank, g.world
assert comm.current() is g and sharding.rank_world() == (r, w)
rows = g.all_gather_i64(np.arange(6, dtype=np.int64) + 100 * r)
assert rows.shape == (w, 6) and all(np.array_equal(rows[k], np.arange(6) + 100 * k) for k in range(w))
objs = g.all_gather_object({"rank": r, "blob": "x" * (1000 * (r + 1))})
assert [o["rank"] for o in objs] == list(range(w)) and len(objs[1]["blob"]) == 2000
assert g.broadcast_object("from0" if r == 0 else None) == "from0"
assert int(g.all_reduce_sum_i64(np.array([r + 1]))[0]) == w * (w + 1) // 2
payload = {k: bytes([k]) * (10 + k) for k in range(5) if k % w == r}
got = sharding.gather_payloads(payload, [k % w for k in range(5)])
if r == 0:
    assert got == [bytes([k]) * (10 + k) for k in range(5)]
else:
    assert got is None
try:
    sharding.agree(ValueError("boom") if r == 1 else None)
    raise SystemExit("agree did not raise")
except ValueError:
    assert r == 1
except RuntimeError as e:
    assert r != 1 and "rank 1 failed: ValueError: boom" in str(e)
g.barrier()
comm.leave()
print("rank", r, "ok")
"""


def test_torch_group_over_gloo_two_ranks(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FTK_ROOT=ROOT, FTK_DIST_TIMEOUT_S="120")
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and f"rank {r} ok" in out, (r, out, err[-2000:])


@pytest.mark.gpu
def test_rccl_communicator_through_the_c_abi_one_rank(tmp_path):
    """ftk_comm_create / ftk_allgather_i64 / ftk_allreduce_sum_i64 / ftk_comm_join / ftk_comm_destroy through ctypes:
    host buffers, device buffers (stream-ordered), the unique id as hex digits and through a rendezvous file."""
    import torch
    from finaletoolkit_amd import _lib as L
    from finaletoolkit_amd.engine import Engine
    lib = L.load()
    with Engine(0) as eng:
        hexid = C.create_string_buffer(257)
        assert lib.ftk_comm_unique_id(hexid) == L.FTK_OK and len(hexid.value) == 256
        for ident in (None, hexid.value, str(tmp_path / "job.id").encode()):
            h = C.c_void_p()
            assert lib.ftk_comm_create(eng.ctx, 0, 1, ident, C.byref(h)) == L.FTK_OK, lib.ftk_last_error(eng.ctx)
            r, w = C.c_int(-1), C.c_int(-1)
            assert lib.ftk_comm_size(h, C.byref(r), C.byref(w)) == L.FTK_OK and (r.value, w.value) == (0, 1)
            if ident is not None and len(ident) != 256:
                # rank 0 wrote the id for the others and took the file away again once the communicator was up (every
                # rank has read it by then): nothing is left for a later job to trip over
                assert not os.path.exists(ident.decode()) and not os.listdir(os.path.dirname(ident.decode()))
            send = np.arange(1000, dtype=np.int64) * 3
            recv = np.zeros(1000, np.int64)
            assert lib.ftk_allgather_i64(h, L.ptr(send), 1000, L.ptr(recv)) == L.FTK_OK
            assert np.array_equal(recv, send)
            v = np.array([5, -7, 1 << 40], np.int64)
            assert lib.ftk_allreduce_sum_i64(h, L.ptr(v), 3) == L.FTK_OK and v.tolist() == [5, -7, 1 << 40]
            d_send = torch.arange(4096, dtype=torch.int64, device="cuda:0")
            d_recv = torch.zeros(4096, dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize()
            assert lib.ftk_allgather_i64(h, L.ptr(d_send), 4096, L.ptr(d_recv)) == L.FTK_OK
            assert lib.ftk_comm_join(h) == L.FTK_OK
            eng.sync()
            assert torch.equal(d_recv, d_send)
            assert lib.ftk_comm_send(h, 0, L.ptr(send), 8) == L.FTK_ERR_INVALID  # nobody else to send to
            lib.ftk_comm_destroy(h)
            if ident is not None and len(ident) != 256:
                assert not os.path.exists(ident.decode())  # rank 0 takes the rendezvous file with it
        h = C.c_void_p()
        assert lib.ftk_comm_create(eng.ctx, 2, 2, hexid.value, C.byref(h)) == L.FTK_ERR_INVALID


@pytest.mark.gpu
def test_a_dead_jobs_rendezvous_file_is_not_joined(tmp_path, monkeypatch):
    """A rank that is not rank 0 ignores a rendezvous file another launch left behind - one that is older than this
    process, one that carries another launch's nonce, one that is a symbolic link - and says so after its time limit
    instead of joining a communicator nobody else is in (ncclCommInitRank would wait for ever)."""
    import subprocess
    code = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ["FTK_ROOT"])
from finaletoolkit_amd import _lib as L
from finaletoolkit_amd.engine import Engine
lib = L.load()
path, kind = sys.argv[1], sys.argv[2]
hexid = C.create_string_buffer(257)
assert lib.ftk_comm_unique_id(hexid) == L.FTK_OK
body = hexid.value.decode() + "\n" + ("other-launch" if kind == "nonce" else os.environ.get("FTK_COMM_NONCE", ""))
if kind == "link":
    open(path + ".real", "w").write(body)
    os.symlink(path + ".real", path)
else:
    open(path, "w").write(body)
if kind == "old":
    os.utime(path, (time.time() - 3600, time.time() - 3600))
with Engine(0) as eng:
    h = C.c_void_p()
    t0 = time.time()
    rc = lib.ftk_comm_create(eng.ctx, 1, 2, path.encode(), C.byref(h))
    print(rc, round(time.time() - t0, 1), lib.ftk_last_error(eng.ctx).decode())
'''
    for kind in ("old", "nonce", "link"):
        env = dict(os.environ, FTK_ROOT=ROOT, FTK_COMM_TIMEOUT_S="1.5", FTK_COMM_NONCE="this-launch")
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / f"{kind}.id"), kind], env=env, capture_output=True, text=True, timeout=300)
        rc, took, msg = r.stdout.strip().split(" ", 2)
        assert r.returncode == 0 and int(rc) == -5 and 1.0 <= float(took) < 20 and "no communicator id of this launch" in msg, (kind, r.stdout, r.stderr[-500:])


@pytest.mark.gpu
def test_two_ranks_over_rccl_on_two_gpus(tmp_path):
    """World 2 over RCCL proper (boxes with two GPUs; skipped on one): the rendezvous through launch_ranks' file, the
    int64 collectives, send / recv and the payload gather of the sharded commands."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: RCCL across ranks needs two")
    from finaletoolkit_amd import sharding
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["FTK_ROOT"])
from finaletoolkit_amd import comm, sharding
g = comm.join("rccl")
r, w = g.rank, g.world
rows = g.all_gather_i64(np.arange(6, dtype=np.int64) + 100 * r)
assert rows.shape == (w, 6) and all(np.array_equal(rows[k], np.arange(6) + 100 * k) for k in range(w))
assert int(g.all_reduce_sum_i64(np.array([r + 1]))[0]) == 3
objs = g.all_gather_object({"rank": r, "blob": "x" * (1000 * (r + 1))})
assert [o["rank"] for o in objs] == [0, 1]
got = sharding.gather_payloads({k: bytes([k]) * (k + 1) for k in range(4) if k % 2 == r}, [0, 1, 0, 1])
assert r != 0 or [bytes(x) for x in got] == [bytes([k]) * (k + 1) for k in range(4)]
comm.leave()
open(os.environ["FTK_OK_FILE"] + str(r), "w").write("ok")
'''
    os.environ["FTK_ROOT"], os.environ["FTK_OK_FILE"] = ROOT, str(tmp_path / "ok")
    try:
        assert sharding.launch_ranks([sys.executable, "-c", code], 2) == 0
    finally:
        os.environ.pop("FTK_OK_FILE", None)
    assert all(os.path.exists(str(tmp_path / "ok") + str(r)) for r in (0, 1))


@pytest.mark.gpu
def test_rccl_group_objects_and_sharding_helpers_one_rank():
    from finaletoolkit_amd import comm, sharding
    g = comm.RcclGroup(0, 1, 0, None)
    try:
        assert (g.rank, g.world, g.backend) == (0, 1, "rccl")
        assert g.all_gather_object({"k": [1, 2, 3]}) == [{"k": [1, 2, 3]}]
        assert g.broadcast_object("x") == "x"
        g.barrier()
        rows = g.all_gather_i64(np.arange(12).reshape(4, 3))
        assert rows.shape == (1, 12)
        assert int(g.all_reduce_sum_i64(np.array([123]))[0]) == 123
        # the sharding helpers on an explicit group
        got = sharding.gather_bin_vectors({"a": np.arange(6).reshape(3, 2)}, ["a"], {"a": 3}, {"a": 1.0}, group=g)
        assert np.array_equal(got["a"], np.arange(6).reshape(3, 2))
    finally:
        g.close()
