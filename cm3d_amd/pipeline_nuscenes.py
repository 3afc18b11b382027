"""nuScenes entry point of the lifting path: what `python 2d_to_3d.py` does in the reference
(src/nuscenes/2d_to_3d.py:343-938), with the per-frame work on the GPU.

Zero-argument invocation uses the reference's module constants (:55-59) and split (`mini_val`, :387);
everything is overridable by flags / environment for tests and other splits.  Under
`torch.distributed.run` scenes are sharded over the ranks and rank 0 writes the single output file
after one gather of box records.
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import dist as cdist
from . import lifting, nusc_io

# module constants of the reference (:55-59)
VER_NAME = "v1.0-trainval"
INPUT_PATH = "../../data/nuScenes/"
OUTPUT_DIR = "../../outputs/nuscenes/"
INPUT_DIR = "../../mask_outputs/nuscenes-detic/"
MINI_VAL = ["scene-0103", "scene-0916"]          # nuscenes.utils.splits.mini_val (:43,:387)
OUTPUT_NAME = "pseudolabels_minival.json"        # :929

META = {"use_camera": True, "use_lidar": False, "use_radar": False, "use_map": True, "use_external": False}   # :357-364


def _load_priors(path):
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)            # cfg/shape_priors_chatgpt.json (:385)
    return dict(lifting.SHAPE_PRIORS_CHATGPT)


_WORKER_TABLES = {}


def prepare_scene_batch(task):
    """Host side of one batch of scenes: reads the frames' files, decodes the RLE strings and packs one HostBatch per mask
    size.  Pure numpy -- never touches the GPU -- so `lift_scenes(workers=N)` can run it in N reader processes.
    task = (version, dataroot, mask_dir, scene names, n_sweeps, ratio, missing_ok, shape priors[, through shared memory]).
    Returns (sample tokens in order, [HostBatch, ...], seconds spent)."""
    version, dataroot, mask_dir, names, n_sweeps, ratio, missing_ok, priors = task[:8]
    t0 = time.time()
    key = (version, dataroot)
    if key not in _WORKER_TABLES:
        _WORKER_TABLES[key] = nusc_io.NuscTables(version, dataroot)
    tables = _WORKER_TABLES[key]
    classes = lifting.ClassTable.nuscenes(priors)
    frames, lanes, frame_lane = [], [], []
    for k, name in enumerate(names):
        scene = tables.scene_by_name(name)
        fs = nusc_io.frames_of_scene(tables, scene, mask_dir, n_sweeps=n_sweeps, ratio=ratio, missing_ok=missing_ok)
        lanes.append(nusc_io.load_lane_points(tables.dataroot, tables.location(scene)))
        frames.extend(fs)
        frame_lane.extend([k] * len(fs))
    tokens = [f.token for f in frames]
    live = [i for i, f in enumerate(frames) if len(f.rles) > 0]
    batches = []
    # one engine call per mask size (all cameras of nuScenes share one)
    for (W, H) in sorted({(frames[i].width, frames[i].height) for i in live}):
        sel = [i for i in live if (frames[i].width, frames[i].height) == (W, H)]
        batches.append(lifting.pack_frames([frames[i] for i in sel], lanes, [frame_lane[i] for i in sel], classes))
    if len(task) > 8 and task[8]:           # reader process: the sweeps (nearly all of the bytes) go through shared memory
        from multiprocessing import shared_memory
        for hb in batches:
            shm = shared_memory.SharedMemory(create=True, size=max(hb.raw.nbytes, 16))
            np.ndarray(hb.raw.shape, np.float32, buffer=shm.buf)[...] = hb.raw
            hb.raw = (shm.name, hb.raw.shape)
            shm.close()
    return tokens, batches, time.time() - t0


def _attach_raw(hb, keep):
    """Maps the shared-memory sweeps a reader process left behind (zero copy); `keep` collects the segments until the
    batch has been uploaded."""
    if isinstance(hb.raw, tuple):
        from multiprocessing import shared_memory
        name, shape = hb.raw
        shm = shared_memory.SharedMemory(name=name)
        keep.append(shm)
        hb.raw = np.ndarray(shape, np.float32, buffer=shm.buf)


def _release(keep):
    for shm in keep:
        try:
            shm.close()
            shm.unlink()
        except (FileNotFoundError, BufferError):
            pass
    keep.clear()


def lift_scenes(tables, scene_names, mask_dir, classes, device, n_sweeps=3, ratio=0.64, masks="rle", timer=None,
                scenes_per_batch=4, missing_ok=False, workers=0, priors=None):
    """Runs the hot path over the given scenes; returns {sample_token: [box dict, ...]} in sample order.
    workers > 0: that many reader processes prepare the batches (prepare_scene_batch: file reads, RLE strings,
    packing -- single-threaded Python does ~360 frames/s of it, the GPU loop 63 k) while this process only uploads,
    launches and collects; each reader loads the tables itself."""
    timer = timer if timer is not None else {}
    results = {}
    pipe = lifting.LiftPipeline(device, depth=2, classes=classes)      # batch i+1 is uploaded while batch i runs
    pending = []                                                       # slots in flight, oldest first

    def drain(keep):
        while len(pending) > keep:
            t1 = time.time()
            hb, res = pipe.collect(pending.pop(0), full=False)      # box records need the per-mask results only
            timer["gpu lifting"] = timer.get("gpu lifting", 0.0) + time.time() - t1
            results.update(lifting.box_records(hb, res, classes))

    _WORKER_TABLES.setdefault((tables.version, tables.dataroot), tables)
    tasks = [(tables.version, tables.dataroot, mask_dir, list(scene_names[b0:b0 + scenes_per_batch]), n_sweeps, ratio, missing_ok, priors)
             for b0 in range(0, len(scene_names), scenes_per_batch)]
    pool, segments = None, []
    if workers > 0 and len(tasks) > 1:
        import multiprocessing as mp
        pool = mp.get_context("spawn").Pool(min(workers, len(tasks)))       # spawn: the readers never inherit GPU state
        prepared = pool.imap(prepare_scene_batch, [t + (True,) for t in tasks])
    else:
        prepared = map(prepare_scene_batch, tasks)
    try:
        for tokens, batches, io_s in prepared:
            timer["io"] = timer.get("io", 0.0) + io_s
            # frames without any mask produce no box but still own a key in the output (:735)
            for tok in tokens:
                results[tok] = []
            for hb in batches:
                t1 = time.time()
                _attach_raw(hb, segments)
                drain(pipe.depth - 1)                                  # the slot about to be reused is free
                pending.append(pipe.submit(hb, masks))
                if segments:                                           # the upload has copied the sweeps: free the segment
                    torch.cuda.current_stream(pipe.dev).synchronize()
                    pipe.streams[pending[-1]].synchronize()
                    hb.raw = None
                    _release(segments)
                timer["gpu lifting"] = timer.get("gpu lifting", 0.0) + time.time() - t1
        drain(0)
    finally:
        _release(segments)
        if pool is not None:
            pool.terminate()
            pool.join()
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="CM3D 2D->3D lifting (nuScenes), MI355X path")
    ap.add_argument("--version", default=os.environ.get("CM3D_VER_NAME", VER_NAME))
    ap.add_argument("--dataroot", default=os.environ.get("CM3D_INPUT_PATH", INPUT_PATH))
    ap.add_argument("--mask-dir", default=os.environ.get("CM3D_INPUT_DIR", INPUT_DIR))
    ap.add_argument("--output-dir", default=os.environ.get("CM3D_OUTPUT_DIR", OUTPUT_DIR))
    ap.add_argument("--output-name", default=OUTPUT_NAME)
    ap.add_argument("--scenes", default=os.environ.get("CM3D_SCENES", ""), help="comma separated scene names (default: mini_val)")
    ap.add_argument("--priors", default="cfg/shape_priors_chatgpt.json")
    ap.add_argument("--ratio", type=float, default=0.64)          # :419
    ap.add_argument("--n-sweeps", type=int, default=3)            # :437
    ap.add_argument("--masks", default="rle", choices=["rle", "dense"])
    ap.add_argument("--missing-ok", action="store_true", help="frames without mask files yield no boxes instead of an error")
    ap.add_argument("--scenes-per-batch", type=int, default=4, help="scenes whose frames form one GPU batch")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("CM3D_WORKERS", "0")),
                    help="reader processes that read and pack the batches (0: in this process)")
    args = ap.parse_args(argv)

    total_start = time.time()
    rank, world, local_rank = cdist.init_from_env()
    if os.environ.get("CM3D_SINGLE_DEVICE"):      # rehearsal of the N>1 path on a one-GPU box (with CM3D_DIST_BACKEND=gloo)
        local_rank = 0
    device = f"cuda:{local_rank}"
    timer = {"io": 0.0, "gpu lifting": 0.0, "gather": 0.0, "total": 0.0}
    tables = nusc_io.NuscTables(args.version, args.dataroot)
    names = [s for s in args.scenes.split(",") if s] or [n for n in MINI_VAL]
    known = {s["name"] for s in tables.scenes()}
    if not args.scenes and not all(n in known for n in names):
        names = sorted(known)
    priors = _load_priors(args.priors)
    classes = lifting.ClassTable.nuscenes(priors)

    # scene-aligned sharding: each rank loads only its scenes' lane tables
    sizes = [tables.scene_by_name(n)["nbr_samples"] for n in names]
    lo, hi = cdist.shard_scenes(sizes, world)[rank]
    mine = lift_scenes(tables, names[lo:hi], args.mask_dir, classes, device, args.n_sweeps, args.ratio, args.masks, timer,
                       missing_ok=args.missing_ok, workers=args.workers, priors=priors, scenes_per_batch=max(1, args.scenes_per_batch))

    t0 = time.time()
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        torch.distributed.gather_object(mine, gathered, dst=0)
        if rank != 0:
            return 0
        results = {}
        for part in gathered:
            results.update(part)
    else:
        results = mine
    timer["gather"] = time.time() - t0

    final_predictions = {"meta": dict(META), "results": results}
    os.makedirs(args.output_dir, exist_ok=True)
    with open(os.path.join(args.output_dir, args.output_name), "w") as f:
        json.dump(final_predictions, f)
    print(f"wrote {len(results)} samples.")
    timer["total"] = time.time() - total_start
    for op, v in timer.items():
        print(op, ":\t\t", v)
    return 0
