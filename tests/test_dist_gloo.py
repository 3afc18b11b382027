"""CPU: the N>1 path (frame sharding + the single gather) with world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cm3d_amd import dist as cdist
    r, w, _ = cdist.init_from_env(backend="gloo")
    a, b = cdist.shard_range(11, r, w)
    # each rank "produces" box records for its frames: (k, 10) doubles, k differs per rank
    rec = torch.tensor([[f, r, 0, 0, 0, 0, 0, 0.5, 1, 3] for f in range(a, b) for _ in range(f % 3 + 1)], dtype=torch.float64)
    out = cdist.gather_records(rec, dst=0)
    if r == 0:
        q.put([o.numpy() for o in out])
    else:
        assert out is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gather_records_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == 2
    frames = np.concatenate([g[:, 0] for g in got])
    # every frame 0..10 present, in order, with its own number of records, none of the padding rows
    exp = np.concatenate([[f] * (f % 3 + 1) for f in range(11)])
    assert np.array_equal(frames, exp)
    assert np.all(got[0][:, 1] == 0) and np.all(got[1][:, 1] == 1)
    assert all(np.all(g[:, 9] == 3) for g in got)


def test_single_process_passthrough():
    from cm3d_amd import dist as cdist
    t = torch.zeros(3, 10, dtype=torch.float64)
    out = cdist.gather_records(t)
    assert len(out) == 1 and out[0] is t
