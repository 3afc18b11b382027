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
_WORKER_READER = None


def _batch_manifest(tables, names, mask_dir, n_sweeps, ratio, missing_ok):
    """The table walk of one batch of scenes (no bulk data): frame manifests, lane tables, lane table of every frame."""
    man, lanes, frame_lane = [], [], []
    for k, name in enumerate(names):
        scene = tables.scene_by_name(name)
        ms = nusc_io.scene_manifest(tables, scene, mask_dir, n_sweeps=n_sweeps, ratio=ratio, missing_ok=missing_ok)
        lanes.append(nusc_io.load_lane_points(tables.dataroot, tables.location(scene)))
        man.extend(ms)
        frame_lane.extend([k] * len(ms))
    return man, lanes, frame_lane


def _with_manifest(task):
    """task -> task + (its manifest,): the first of the two host stages when they run on threads of their own (lift_scenes)."""
    version, dataroot, mask_dir, names, n_sweeps, ratio, missing_ok = task[:7]
    t0 = time.time()
    tables = _WORKER_TABLES[(version, dataroot)]
    man = _batch_manifest(tables, names, mask_dir, n_sweeps, ratio, missing_ok)
    return task + (man, time.time() - t0)


def _load_prepared(task):
    """Second host stage: prepare_scene_batch on a task that brings its manifest; adds the first stage's time."""
    tokens, batches, io_s = prepare_scene_batch(task[:11])
    return tokens, batches, io_s + task[11]


def prepare_scene_batch(task):
    """Host side of one batch of scenes: reads the frames' files, decodes the RLE strings and packs one HostBatch per mask
    size.  Pure numpy -- never touches the GPU -- so `lift_scenes(workers=N)` can run it in N reader processes.
    task = (version, dataroot, mask_dir, scene names, n_sweeps, ratio, missing_ok, shape priors[, through shared memory]).
    Returns (sample tokens in order, [HostBatch, ...], seconds spent)."""
    version, dataroot, mask_dir, names, n_sweeps, ratio, missing_ok, priors = task[:8]
    t0 = time.time()
    key = (version, dataroot)
    if key not in _WORKER_TABLES:
        _WORKER_TABLES[key] = nusc_io.NuscTables(version, dataroot, annotations=False)
    tables = _WORKER_TABLES[key]
    classes = lifting.ClassTable.nuscenes(priors)
    native = task[9] if len(task) > 9 else None
    through_shm = len(task) > 8 and task[8]
    if isinstance(native, int):            # a reader process: its own loader with that many threads, created once
        global _WORKER_READER
        if _WORKER_READER is None or _WORKER_READER[0] != native:
            from . import reader as rdmod
            _WORKER_READER = (native, rdmod.Reader(native, pinned=False))
        native = _WORKER_READER[1]
    if native is not None:
        # the native loader (libcm3d_reader.so): this thread only walks the tables and the small json files; sweeps and mask
        # pickles of the whole batch are read and parsed by the loader's thread pool, into page-locked staging buffers
        man, lanes, frame_lane = task[10] if len(task) > 10 else _batch_manifest(tables, names, mask_dir, n_sweeps, ratio, missing_ok)
        from .reader import ERR_FORMAT, ReaderError
        segs = []

        def shm_alloc(n_floats):                # the sweeps go straight from the page cache into the segment the parent maps
            from multiprocessing import shared_memory
            shm = shared_memory.SharedMemory(create=True, size=max(4 * int(n_floats), 16))
            segs.append(shm)
            return np.ndarray((int(n_floats),), np.float32, buffer=shm.buf)
        try:
            hb, _ = lifting.pack_manifest(man, lanes, frame_lane, classes, native, alloc=shm_alloc if through_shm else None)
            if through_shm and hb is not None:
                shape = hb.raw.shape
                hb.raw = (segs[-1].name, shape)
                for shm in segs[:-1]:              # a repack after dropping mask-less frames allocated twice
                    shm.close(); shm.unlink()
                segs[-1].close()
            return [m.token for m in man], ([hb] if hb is not None else []), time.time() - t0
        except ReaderError as exc:
            if exc.code != ERR_FORMAT:
                raise                 # a pickle the native parser does not know: the Python reader below takes the batch
        except ValueError as exc:
            if "share mask size" not in str(exc):
                raise                 # frames with different mask sizes: the Python reader below groups them
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
    if through_shm:                         # reader process: the sweeps (nearly all of the bytes) go through shared memory
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


def _ahead(it, depth=1):
    """Iterates `it` in a background thread, `depth` items ahead.  The native loader releases the interpreter lock while its
    threads read, so the Python half of the next batch (table walk, packing) runs under the file reads of this one."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    end = object()

    def work():
        try:
            for x in it:
                q.put(x)
            q.put(end)
        except BaseException as exc:       # handed to the consumer
            q.put(exc)
    threading.Thread(target=work, daemon=True).start()
    while True:
        x = q.get()
        if x is end:
            return
        if isinstance(x, BaseException):
            raise x
        yield x


def _tokens_as_rows(hb, sample_rows):
    """A fallback batch of the native route: its frames named by their row in sample.json, like the native batches'."""
    hb.tokens = [sample_rows[t] for t in hb.tokens]
    return hb


def sample_tokens(tables, scene_names):
    """The job's sample tokens in output order (scenes in the given order, samples in scene order): every rank derives the
    same list from the tables alone, so a shipped record only needs its index into it."""
    return [s["token"] for name in scene_names for s in tables.samples_of_scene(tables.scene_by_name(name))]


def lift_scenes(tables, scene_names, mask_dir, classes, device, n_sweeps=3, ratio=0.64, masks="rle", timer=None,
                scenes_per_batch=4, missing_ok=False, workers=0, priors=None, token_index=None, reader_threads=-1):
    """Runs the hot path over the given scenes; returns the kept-box records of all their frames as ONE device tensor
    (lifting.kept_box_records; column 5 = token_index[sample token], default: the index into sample_tokens(scene_names)).
    workers > 0: that many reader processes prepare the batches (prepare_scene_batch: file reads, RLE strings,
    packing -- single-threaded Python does ~360 frames/s of it, the GPU loop 63 k) while this process only uploads,
    launches and collects; each reader loads the tables itself."""
    timer = timer if timer is not None else {}
    if token_index is None:
        token_index = {t: i for i, t in enumerate(sample_tokens(tables, scene_names))}
    records = []
    pipe = lifting.LiftPipeline(device, depth=2, classes=classes)      # batch i+1 is uploaded while batch i runs
    pending = []                                                       # (slot, frame ids) in flight, oldest first

    def drain(keep):
        while len(pending) > keep:
            t1 = time.time()
            slot, ids = pending.pop(0)
            records.append(pipe.collect_records(slot, ids))           # stays on the device until the one gather at the end
            timer["gpu lifting"] = timer.get("gpu lifting", 0.0) + time.time() - t1

    _WORKER_TABLES.setdefault((tables.version, tables.dataroot), tables)
    tasks = [(tables.version, tables.dataroot, mask_dir, list(scene_names[b0:b0 + scenes_per_batch]), n_sweeps, ratio, missing_ok, priors)
             for b0 in range(0, len(scene_names), scenes_per_batch)]
    pool, segments = None, []
    if reader_threads >= 0 and workers <= 0:
        from . import reader as rdmod
        rd = rdmod.Reader(reader_threads)              # 0 = one thread per core
        tasks = [t + (False, rd) for t in tasks]
    if workers > 0 and len(tasks) > 1:
        import multiprocessing as mp
        nproc = min(workers, len(tasks))
        pool = mp.get_context("spawn").Pool(nproc)     # spawn: the readers never inherit GPU state
        if reader_threads >= 0:                        # every reader process runs the native loader on its share of the cores
            per = reader_threads if reader_threads > 0 else max(1, (os.cpu_count() or nproc) // nproc)
            prepared = pool.imap(prepare_scene_batch, [t + (True, int(per)) for t in tasks])
        else:
            prepared = pool.imap(prepare_scene_batch, [t + (True,) for t in tasks])
    elif reader_threads >= 0 and len(tasks) > 1:
        # three host threads in a row: the table walk (Python) of batch k+2, the file reads (native threads, no interpreter
        # lock) and packing of batch k+1, and this one, which uploads and launches batch k
        prepared = _ahead(map(_load_prepared, _ahead(map(_with_manifest, tasks))))
    else:
        prepared = map(prepare_scene_batch, tasks)
    try:
        for tokens, batches, io_s in prepared:
            timer["io"] = timer.get("io", 0.0) + io_s
            for hb in batches:
                t1 = time.time()
                _attach_raw(hb, segments)
                drain(pipe.depth - 1)                                  # the slot about to be reused is free
                ids = np.array([[token_index[t], 0] for t in hb.tokens], np.float64)
                pending.append((pipe.submit(hb, masks), ids))
                if segments:                                           # the upload has copied the sweeps: free the segment
                    pipe.uploaded[pending[-1][0]].synchronize()        # (an event behind the H2D copies, not the whole pass)
                    hb.raw = None
                    _release(segments)
                timer["gpu lifting"] = timer.get("gpu lifting", 0.0) + time.time() - t1
        drain(0)
    finally:
        _release(segments)
        if pool is not None:
            pool.terminate()
            pool.join()
    if records:
        return torch.cat(records, 0)
    return torch.zeros(0, 10, dtype=torch.float64, device=device)


REFERENCE_BUCKETS = ("io", "points in mask", "medoid", "drivable", "closest lane", "nms")       # the reference's timer (:368-378)


def _native_batch_lanes(nt, names, lane_cache):
    """The scenes' lane tables (one per map location, cached) and the table every frame of the batch uses."""
    # one table per distinct LOCATION of the batch (four scenes of one city used to upload and index that city's table
    # four times), in sorted order so that batches with the same set of cities carry the same tables -- the engines' shared cache
    # of lane indices (lifting.LiftEngine) is keyed by that set
    scene_loc = [nt.location(name) for name in names]
    locs = sorted(set(scene_loc))
    for loc in locs:
        if loc not in lane_cache:
            lane_cache[loc] = np.asarray(nusc_io.load_lane_points(nt.dataroot, loc), np.float64).astype(np.float32).reshape(-1, 3)   # torch.Tensor(...) at :278
    lanes = [lane_cache[loc] for loc in locs]
    frame_lane = []
    for name, loc in zip(names, scene_loc):
        frame_lane.extend([locs.index(loc)] * nt.scene_samples(name))
    return lanes, frame_lane, locs


def _native_batch_head(nt, names, mask_dir, n_sweeps, ratio, classes, missing_ok, lane_cache, sub=None, pool=None):
    """First host stage of a batch on the native route: table walk, <f>_data.json, mask pickles + RLE strings (reader.Tables /
    reader.Manifest) and the scenes' lane tables -- the latter on `pool` (a one-thread executor) beside the former when one is
    given: a job whose scenes all have their own map location loads a table per scene.  Returns what _native_batch_tail needs, or
    None when the batch needs the Python reader (frames without masks to be dropped, mixed mask sizes, a pickle the native parser
    does not know)."""
    from .reader import ERR_FORMAT, ReaderError
    t = [time.perf_counter()]
    lanes_job = pool.submit(_native_batch_lanes, nt, names, lane_cache) if pool is not None else None
    try:
        man = nt.manifest(names, mask_dir, n_sweeps, ratio, classes.names, missing_ok)
        t.append(time.perf_counter())
        counts, rle_off, fmo, wh = man.load_masks()
        t.append(time.perf_counter())
    except ReaderError as exc:
        if exc.code != ERR_FORMAT:
            raise
        return None
    n_per = np.diff(fmo)
    if man.n_masks == 0 or (n_per == 0).any() or not np.array_equal(fmo, man.frame_mask_off):
        return None
    W, H = int(wh[0, 0]), int(wh[0, 1])
    if (wh[:, 0] != W).any() or (wh[:, 1] != H).any():
        return None
    lanes, frame_lane, locs = lanes_job.result() if lanes_job is not None else _native_batch_lanes(nt, names, lane_cache)
    t.append(time.perf_counter())
    if sub is not None:
        for key, a in (("host table walk + data.json", 0), ("host mask files", 1), ("host lane tables", 2)):
            sub[key] = sub.get(key, 0.0) + t[a + 1] - t[a]
    return man, counts, rle_off, fmo, n_per, (W, H), lanes, frame_lane, locs


def _native_batch_tail(nt, head, sub=None, rd=None):
    """Second host stage: the batch's sweeps read straight into a page-locked buffer, and the HostBatch put together -- without
    a Python statement per frame.  Returns (HostBatch, rows of its frames in sample.json)."""
    man, counts, rle_off, fmo, n_per, (W, H), lanes, frame_lane, locs = head
    t0 = time.perf_counter()
    # the sweeps go straight into the batch's layout: quads (12 of a row's 20 bytes reach the page-locked buffer and the GPU; the
    # intensity is not uploaded -- no output of the entry point holds it) unless CM3D_RAW_LAYOUT=rows asks for the files' rows
    intensity = frame_rows = None
    if lifting.default_layout() == "quads":
        raw, intensity, row_off, frame_rows = man.load_sweeps_quads(5, rd)
        stride = lifting._lib.RAW_QUADS
    else:
        raw, row_off = man.load_sweeps(5, rd)
        stride = 5
    t1 = time.perf_counter()
    F = man.n_frames
    hb = lifting.HostBatch(
        raw=raw, raw_stride=stride, intensity=intensity, frame_rows=frame_rows, sweep_row_off=row_off, sweep_xf=man.sweep_xf, frame_sweep_off=man.frame_sweep_off,
        max_rows_per_sweep=max(1, int(np.diff(row_off).max())), cams=man.cams, n_cams=6, mask_off=fmo.astype(np.int32), mask_cam=man.mask_cam,
        mask_frame=np.repeat(np.arange(F, dtype=np.int32), n_per), rle_counts=counts, rle_off=rle_off, class_id=man.class_id, score=man.score,
        lane=np.concatenate(lanes, 0), lane_off=np.concatenate([[0], np.cumsum([t_.shape[0] for t_ in lanes])]).astype(np.int32),
        frame_lane=np.asarray(frame_lane, np.int32), ego_xyz=man.ego_xyz, width=W, height=H, tokens=[None] * F, labels=None, ego_box=True)
    hb.lane_key = (nt.dataroot, tuple(locs))              # which lane tables these are: LiftEngine keeps their spatial index across batches
    if sub is not None:
        sub["host sweep files"] = sub.get("host sweep files", 0.0) + t1 - t0
        sub["host assemble"] = sub.get("host assemble", 0.0) + time.perf_counter() - t1
    return hb, man.sample_index


def _native_batch(nt, names, mask_dir, n_sweeps, ratio, classes, missing_ok, lane_cache, sub=None):
    head = _native_batch_head(nt, names, mask_dir, n_sweeps, ratio, classes, missing_ok, lane_cache, sub)
    return None if head is None else _native_batch_tail(nt, head, sub)


def lift_scenes_native(nt, scene_names, mask_dir, classes, device, row_to_out, n_sweeps=3, ratio=0.64, masks="rle", timer=None,
                       scenes_per_batch=4, missing_ok=False, python_batch=None, on_records=None):
    """lift_scenes with the whole host side in libcm3d_reader.so (reader.Tables): per batch one native table walk, one native read
    of all its files, one upload, one pass.  Three host threads in a row -- the reads of batch k+1 run under the upload and the
    launches of batch k -- and two batches in flight on the GPU.  row_to_out: row in sample.json -> index of the sample in the
    job's output order.  python_batch(names) -> (tokens, [HostBatch]) is the fallback for a batch the native path declines.
    Per-stage GPU times go into `timer` under the reference's bucket names.  on_records(records of one batch as numpy, first
    output index, number of samples): called per finished batch whose samples are a contiguous run of the output order, with
    (None, 0, 0) otherwise -- the caller can then write the batch's share of the result file while later batches still run."""
    timer = timer if timer is not None else {}
    records, pending, stage_ev = [], [], []
    pipe = lifting.LiftPipeline(device, depth=2, classes=classes)
    lane_cache = {}
    from . import reader as rdmod
    rd_sweeps = rdmod.Reader(nt.rd.threads)                    # the sweep stage runs on a thread of its own: its own pool and buffers
    # (both pools together oversubscribe the cores by two: the table / mask stage is short and mostly waits on the sweep stage)

    def drain(keep):
        while len(pending) > keep:
            t1 = time.time()
            slot, ids, evs = pending.pop(0)
            records.append(pipe.collect_records(slot, ids))
            timer["gpu lifting"] = timer.get("gpu lifting", 0.0) + time.time() - t1
            if on_records is not None:
                first = int(ids[0, 0]) if len(ids) else 0
                if len(ids) and np.array_equal(ids[:, 0], np.arange(first, first + len(ids))):
                    on_records(records[-1].cpu().numpy(), first, len(ids))
                else:
                    on_records(None, 0, 0)
            if len(evs) == 5:                                   # (complete: the slot's stream has been synchronised)
                for key, a, b in (("points in mask", 0, 1), ("medoid", 1, 2), ("closest lane", 2, 3), ("nms", 3, 4)):
                    timer[key] = timer.get(key, 0.0) + evs[a].elapsed_time(evs[b]) * 1e-3

    # three host threads in a row: (1) table walk, data files, mask files and lane tables of batch k+2, (2) the sweep files of
    # batch k+1 into a page-locked buffer, (3) this one: upload and launches of batch k.  The native calls release the
    # interpreter lock, so the three really overlap.
    from concurrent.futures import ThreadPoolExecutor
    lane_pool = ThreadPoolExecutor(max_workers=1)

    def heads():
        for b0 in range(0, len(scene_names), scenes_per_batch):
            names = list(scene_names[b0:b0 + scenes_per_batch])
            t0 = time.time()
            head = _native_batch_head(nt, names, mask_dir, n_sweeps, ratio, classes, missing_ok, lane_cache, timer, lane_pool)
            yield names, head, time.time() - t0

    def prepared():
        for names, head, io_s in _ahead(heads(), depth=1):
            t0 = time.time()
            got = None if head is None else _native_batch_tail(nt, head, timer, rd_sweeps)
            yield names, got, max(io_s, time.time() - t0)       # the two stages overlap: the longer one is the wall time

    for names, got, io_s in _ahead(prepared(), depth=1):
        timer["io"] = timer.get("io", 0.0) + io_s
        if got is None:                                         # the Python reader takes this batch
            t0 = time.time()
            tokens, batches = python_batch(names)
            timer["io"] += time.time() - t0
            todo = [(hb, None) for hb in batches]
        else:
            todo = [got]
        for hb, rows in todo:
            t1 = time.time()
            drain(pipe.depth - 1)
            if rows is None:
                ids = np.array([[row_to_out[t], 0] for t in hb.tokens], np.float64)
            else:
                ids = np.stack([row_to_out[rows].astype(np.float64), np.zeros(len(rows))], 1)
            evs = []
            t2 = time.time()
            pending.append((pipe.submit(hb, masks, stage_events=evs), ids, evs))
            timer["host upload + launches"] = timer.get("host upload + launches", 0.0) + time.time() - t2
            timer["gpu lifting"] = timer.get("gpu lifting", 0.0) + time.time() - t1
    drain(0)
    lane_pool.shutdown(wait=False)
    if records:
        return torch.cat(records, 0)
    return torch.zeros(0, 10, dtype=torch.float64, device=device)


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
                    help="reader processes that read and pack the batches with the Python reader (0: in this process)")
    ap.add_argument("--reader-threads", type=int, default=int(os.environ.get("CM3D_READER_THREADS", "0")),
                    help="threads of the native loader (libcm3d_reader.so) that reads sweeps and mask files; 0 = one per core, "
                         "-1 = the Python reader (pickle.load / np.fromfile per frame, like the reference)")
    args = ap.parse_args(argv)

    total_start = time.time()
    rank, world, local_rank = cdist.init_from_env()
    if os.environ.get("CM3D_SINGLE_DEVICE"):      # rehearsal of the N>1 path on a one-GPU box (with CM3D_DIST_BACKEND=gloo)
        local_rank = 0
    device = f"cuda:{local_rank}"
    timer = {k: 0.0 for k in REFERENCE_BUCKETS}
    timer.update({"tables": 0.0, "gpu lifting": 0.0, "gather": 0.0, "write": 0.0, "total": 0.0})
    priors = _load_priors(args.priors)
    classes = lifting.ClassTable.nuscenes(priors)
    native = args.reader_threads >= 0 and args.workers <= 0
    py_tables = []

    def tables_py():                       # the Python tables, when something needs them (fallback batches, the Python reader)
        if not py_tables:
            py_tables.append(nusc_io.NuscTables(args.version, args.dataroot, annotations=False))
        return py_tables[0]
    if native:
        # the whole host side in libcm3d_reader.so: tables parsed once (natively, on the reader's threads), per batch one native
        # table walk + one native read of all its files; the Python tables are only loaded if a batch falls back
        from . import reader as rdmod
        rd = rdmod.Reader(args.reader_threads)
        nt = rdmod.Tables(rd, args.dataroot, args.version)
        timer["tables"] = time.time() - total_start
        known = nt.scene_names()
        names = [s for s in args.scenes.split(",") if s] or [n for n in MINI_VAL]
        if not args.scenes and not all(n in known for n in names):
            names = sorted(known)
        sizes = [nt.scene_samples(n) for n in names]
        lo, hi = cdist.shard_scenes(sizes, world)[rank]
        tokens, rows = nt.job_tokens(names)                    # the whole job's samples, identical on every rank
        row_to_out = np.full(int(rows.max()) + 1 if len(rows) else 1, -1, np.int64)
        row_to_out[rows] = np.arange(len(rows))

        def python_batch(batch_names):
            tabs = tables_py()
            toks, batches, _ = prepare_scene_batch((args.version, args.dataroot, args.mask_dir, batch_names, args.n_sweeps, args.ratio,
                                                    args.missing_ok, priors))
            sample_rows = {t: i for i, t in enumerate(tabs.t["sample"].keys())}
            return toks, [_tokens_as_rows(hb, sample_rows) for hb in batches]
        # one rank: every finished batch's share of the result file is formatted right away, under the GPU work of the next ones
        # (on a thread of its own: formatting a batch's boxes takes as long as uploading and launching the next batch)
        import queue
        import threading
        parts, next_first, streamed = [], [0], [world == 1]
        write_q = queue.Queue()

        def write_worker():
            while True:
                item = write_q.get()
                if item is None:
                    return
                rec_np, first, count = item
                t0 = time.time()
                if rec_np is None or first != next_first[0]:
                    streamed[0] = False
                elif streamed[0]:
                    parts.append(lifting.nuscenes_results_json_native(rec_np, tokens, classes, None, part=(first, count)))
                    next_first[0] = first + count
                timer["write"] += time.time() - t0
        writer = threading.Thread(target=write_worker, daemon=True)
        writer.start()

        def on_records(rec_np, first, count):
            write_q.put((rec_np, first, count))
        mine = lift_scenes_native(nt, names[lo:hi], args.mask_dir, classes, device, row_to_out, args.n_sweeps, args.ratio, args.masks, timer,
                                  scenes_per_batch=max(1, args.scenes_per_batch), missing_ok=args.missing_ok, python_batch=python_batch,
                                  on_records=on_records if world == 1 else None)
        write_q.put(None)
        writer.join()
        streamed[0] = streamed[0] and next_first[0] == len(tokens)
    else:
        tables = tables_py()
        timer["tables"] = time.time() - total_start
        names = [s for s in args.scenes.split(",") if s] or [n for n in MINI_VAL]
        known = {s["name"] for s in tables.scenes()}
        if not args.scenes and not all(n in known for n in names):
            names = sorted(known)
        # scene-aligned sharding: each rank loads only its scenes' lane tables
        sizes = [tables.scene_by_name(n)["nbr_samples"] for n in names]
        lo, hi = cdist.shard_scenes(sizes, world)[rank]
        tokens = sample_tokens(tables, names)                  # the whole job's samples, identical on every rank
        mine = lift_scenes(tables, names[lo:hi], args.mask_dir, classes, device, args.n_sweeps, args.ratio, args.masks, timer,
                           missing_ok=args.missing_ok, workers=args.workers, priors=priors, scenes_per_batch=max(1, args.scenes_per_batch),
                           token_index={t: i for i, t in enumerate(tokens)}, reader_threads=args.reader_threads)

    # the single exchange of the job: fixed-size box records -> rank 0 (one RCCL all_gather of payload; also the path of a
    # one-rank run).  Ranks hold contiguous scene blocks, so rank order is sample order.
    t0 = time.time()
    gathered = cdist.gather_records(mine, dst=0)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return 0
    rec = torch.cat([g.cpu() for g in gathered], 0).numpy()
    timer["gather"] = time.time() - t0

    # the writer (:929-930): the records straight to the text json.dump would produce for the reference's dict of box lists
    t0 = time.time()
    os.makedirs(args.output_dir, exist_ok=True)
    if native:
        with open(os.path.join(args.output_dir, args.output_name), "wb") as f:
            if streamed[0]:
                f.write(lifting.nuscenes_results_json_head(dict(META)) + b", ".join(parts) + b"}}")
            else:
                f.write(lifting.nuscenes_results_json_native(rec, tokens, classes, dict(META)))
    else:
        text, _ = lifting.nuscenes_results_json(rec, tokens, classes, dict(META))
        with open(os.path.join(args.output_dir, args.output_name), "w") as f:
            f.write(text)
    timer["write"] += time.time() - t0
    print(f"wrote {len(tokens)} samples.")
    timer["total"] = time.time() - total_start
    for op, v in timer.items():
        print(op, ":\t\t", v)
    return 0
