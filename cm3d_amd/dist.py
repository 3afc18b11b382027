"""Frame sharding across the GPUs of one node and the single end-of-job gather.

The reference is a single process (no torch.distributed anywhere); frames are
independent (stage 1 per (frame, mask), stage 2 per box, NMS per frame --
src/nuscenes/2d_to_3d.py:415-694, :733-822, :844-924), so the only exchange is
one gather of fixed-size box records to rank 0 (`gather_records`): an `all_gather` of the
per-rank record counts, then one padded `all_gather` of the records over RCCL (backend "nccl" on
ROCm) or gloo on CPU; rank 0 keeps the result.  The entry points (pipeline_nuscenes, pipeline_waymo)
and bench.py all use this one exchange -- also in a one-rank run, where it is a pass-through.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = os.environ.get("CM3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        import datetime
        # a rank that never arrives (a crashed neighbour, a wrong WORLD_SIZE) must end the job with an error, not hang it: every
        # collective of this job is small, so a short timeout is safe (CM3D_DIST_TIMEOUT_S overrides)
        timeout = datetime.timedelta(seconds=float(os.environ.get("CM3D_DIST_TIMEOUT_S", "300")))
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            # device_id: RCCL binds to this rank's GPU and builds its communicator here, eagerly, instead of inside the first
            # collective (which for this job would be the end-of-run gather, inside the timed region of bench.py)
            kw["device_id"] = torch.device(f"cuda:{local_rank}")
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout, **kw)
        except TypeError:               # a torch whose init_process_group knows no device_id
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local_rank


class GatherError(RuntimeError):
    """The end-of-job exchange failed (a rank missing, a timeout): the job's result is incomplete."""


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block of items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_scenes(scene_sizes: List[int], world: int):
    """Assign whole scenes to ranks (a rank then only needs its own scenes' lane tables), in
    scene order, balancing frame counts.  Returns (first_scene, last_scene_exclusive) per rank;
    ranks beyond the number of scenes get an empty range."""
    n, total = len(scene_sizes), sum(scene_sizes)
    bounds, s, acc = [], 0, 0
    for r in range(world):
        first = s
        if r == world - 1:
            s = n
        else:
            ranks_after = world - r - 1
            target = total * (r + 1) / world
            while s < n and (s == first or ((n - s) > ranks_after and acc + scene_sizes[s] / 2.0 <= target)):
                acc += scene_sizes[s]
                s += 1
        bounds.append((first, s))
    return bounds


def gather_records(records: torch.Tensor, dst: int = 0):
    """records: (k, R) tensor on this rank (k may differ per rank).  Returns on `dst` the list of
    per-rank (k_r, R) tensors in rank order, elsewhere None.  One collective of payload."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [records]
    world, rank = dist.get_world_size(), dist.get_rank()
    try:
        k = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
        counts = [torch.zeros_like(k) for _ in range(world)]
        dist.all_gather(counts, k)
        counts = [int(c.item()) for c in counts]
        kmax = max(counts)
        padded = records
        if records.shape[0] < kmax:
            pad = torch.zeros(kmax - records.shape[0], records.shape[1], dtype=records.dtype, device=records.device)
            padded = torch.cat([records, pad], 0)
        padded = padded.contiguous()
        # all_gather rather than gather: the same single exchange (5120 x 80 B per rank on the C2 batch), and the collective
        # every backend of torch.distributed implements for device tensors
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded)
        if padded.is_cuda:
            torch.cuda.synchronize(padded.device)       # an asynchronous RCCL error or timeout surfaces here, not later
    except Exception as exc:                             # (DistBackendError, a timeout, a peer that left)
        raise GatherError(f"rank {rank} of {world}: the end-of-job gather of the box records failed: {exc!r}") from exc
    if rank != dst:
        return None
    return [b[:c] for b, c in zip(bufs, counts)]
