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


def _c3_scene_sizes():
    """850 scenes, 28 130 key frames (nuScenes trainval, BASELINE config C3): about 28..40 frames per scene, seeded."""
    rng = np.random.default_rng(3)
    sizes = rng.integers(28, 39, 850)
    while sizes.sum() != 28130:                      # spread the remainder over the scenes
        i = int(rng.integers(0, 850))
        sizes[i] += 1 if sizes.sum() < 28130 else -1
    assert sizes.min() > 20 and sizes.sum() == 28130
    return sizes.tolist()


def _worker_c3(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    from cm3d_amd import dist as cdist
    r, w, _ = cdist.init_from_env(backend="gloo")
    sizes = _c3_scene_sizes()
    first = np.concatenate([[0], np.cumsum(sizes)])
    s0, s1 = cdist.shard_scenes(sizes, w)[r]
    # the records a rank ships: one row of 10 doubles per kept box, (global frame number, rank) in the identity columns
    # (lifting.kept_box_records); 0..3 boxes per frame, so some frames ship nothing at all
    frames = np.arange(first[s0], first[s1])
    per = frames % 4
    rows = np.repeat(frames, per)
    rec = torch.zeros(rows.size, 10, dtype=torch.float64)
    rec[:, 5] = torch.from_numpy(rows.astype(np.float64))
    rec[:, 6] = float(r)
    rec[:, 9] = 3.0
    out = cdist.gather_records(rec, dst=0)
    if r == 0:
        q.put([(o[:, 5].numpy().astype(np.int64), o[:, 6].numpy().astype(np.int64), float(o[:, 9].min()) if o.shape[0] else 3.0) for o in out])
    else:
        assert out is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gather_records_eight_ranks_c3_shape():
    """BASELINE C3 at the real rank count, on the CPU: 28 130 frames of 850 scenes, scene-aligned blocks over 8 ranks
    (dist.shard_scenes), every rank's kept-box records through the ONE exchange; rank 0 sees every frame's records once, in
    frame order, from the rank that owns the frame's scene, and no padding row."""
    from cm3d_amd import dist as cdist
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c3, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert len(got) == world
    sizes = _c3_scene_sizes()
    first = np.concatenate([[0], np.cumsum(sizes)])
    bounds = cdist.shard_scenes(sizes, world)
    frames = np.concatenate([g[0] for g in got])
    exp = np.repeat(np.arange(28130), np.arange(28130) % 4)
    assert np.array_equal(frames, exp)                                   # global frame order, nothing lost, nothing doubled
    for r, (fr, rk, flag) in enumerate(got):
        assert np.all(rk == r) and flag == 3.0                           # each block from its own rank; no zero padding row
        a, b = first[bounds[r][0]], first[bounds[r][1]]
        assert fr.size == 0 or (fr.min() >= a and fr.max() < b)          # scene-aligned: only frames of the rank's own scenes
    per_rank = [int(first[b] - first[a]) for a, b in bounds]
    assert max(per_rank) - min(per_rank) <= 2 * 40                       # balanced to within a scene at either end


def _worker_missing_peer(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CM3D_DIST_TIMEOUT_S="5")
    from cm3d_amd import dist as cdist
    r, w, _ = cdist.init_from_env(backend="gloo")
    if r == 1:
        os._exit(0)                              # this rank dies before the exchange
    try:
        cdist.gather_records(torch.zeros(3, 10, dtype=torch.float64), dst=0)
        q.put("no error")
    except cdist.GatherError as exc:
        q.put("GatherError: " + str(exc)[:60])
    q.close()
    q.join_thread()                              # (the queue's feeder thread must have written the item before the process ends)
    os._exit(0)


def test_a_missing_rank_ends_the_gather_with_an_error_not_a_hang():
    """The one exchange has a timeout and a clear error: rank 1 leaves before it, rank 0 must get GatherError within
    the timeout (5 s here) instead of waiting for ever."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_missing_peer, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
    assert got.startswith("GatherError"), got
