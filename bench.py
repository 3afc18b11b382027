#!/usr/bin/env python3
"""Benchmark of the lifting hot path (BASELINE.json: pseudo-label frames/sec on
nuScenes-shaped sweeps).

  python bench.py --gpus N --steps K --warmup W
  N>1 without RANK in the environment: bench.py starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
  --master-addr 127.0.0.1 ... bench.py <same flags>` itself, as a child process and before anything touches the GPU,
  relays rank 0's JSON line and exits with the child's code.  Under torchrun (RANK set) WORLD_SIZE must equal --gpus.

A step = one pass of the whole path (a2 sweep prep, a1+a3 mask expansion/erosion, a4-a8
projection + in-mask gather + ordered compaction, a9 medoid, a10 lane NN, a11-a15 boxes + NMS)
over one batch of synthetic frames resident in HBM.  Default workload = BASELINE config C2:
256 frames x (35k points, 6 cameras, 20 masks of 1600x900).  Every rank owns its own batch
(weak scaling); after the K steps one RCCL gather ships the fixed-size box records to rank 0.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import platform
import sys
import time

# Four batches in flight want four hardware queues of their own and the runtime maps streams onto GPU_MAX_HW_QUEUES of them (4 by default, one
# of which other work shares): read when the HIP runtime starts, so it is set before anything can start it.  (A value in the environment wins.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cm3d_amd import dist as cdist, lifting, synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def packed_rect_bytes(bbox, Wp):
    """Bytes of the bit-packed masks that exist at all: the RLE path writes, and the projection can only read, the words
    of each eroded mask's bounding rectangle.  bbox: (M,4) int32 x0,y0,x1,y1 inclusive, x0>x1 when empty."""
    b = np.asarray(bbox, np.int64)
    ok = (b[:, 2] >= b[:, 0]) & (b[:, 3] >= b[:, 1])
    words = np.where(ok, ((b[:, 2] >> 5) - (b[:, 0] >> 5) + 1) * (b[:, 3] - b[:, 1] + 1), 0)
    return int(4 * words.sum())


def compulsory_bytes(hb, planes, sum_hits, mode, fused, cloud_stored, rect_bytes, hit_rows=None):
    """Bytes every launch MUST move across HBM (per-unit figure x units of one launch).  Nothing here is a
    byte the kernels skip: the full bit-packed masks of SURVEY 8(d) (n*ceil(W*H/8) per frame) are never written nor read
    as a whole -- only `rect_bytes` of them exist -- so they are NOT part of any figure called achieved / frac."""
    F, M = hb.n_frames, hb.n_masks
    W, H = hb.width, hb.height
    rows = hb.n_raw_rows
    raw = hb.bytes_per_row * rows
    cloud = 16 * rows
    # hit words exist only for the 256-row blocks that hold an in-mask point (a third of them on the headline shape): the
    # projection writes, and the compaction reads, those and no others (r03; before: every row's)
    hitw = 4 * (rows if hit_rows is None else hit_rows) * planes
    runs = 4 * int(hb.rle_counts.size)
    by = {
        # projection launch: raw rows in (fused) or the prepared cloud in, hit words out, cloud out when it is kept.
        # The mask words a launch fetches (a subset of rect_bytes, mostly L2 hits) are extra traffic, not counted.
        "k_project_hits": (raw if fused else cloud) + hitw + (cloud if (fused and cloud_stored) else 0),
        "k_sweep_xform": raw + cloud,
        "k_rle_erode_pack_wave": runs + rect_bytes,
        "k_erode_pack": M * W * H + M * ((W + 31) // 32) * 4 * H,           # dense path: really streams both
        "k_compact_hits": hitw + 8 * sum_hits + (12 * sum_hits if not cloud_stored else 0),
        "k_medoid": 16 * sum_hits,
        "boxes": 80 * M,
    }
    masks = by["k_erode_pack"] if mode == "dense" else by["k_rle_erode_pack_wave"]
    by["pass_total"] = (by["k_project_hits"] + (0 if fused else by["k_sweep_xform"]) + masks + by["k_compact_hits"] +
                        by["k_medoid"] + by["boxes"])
    return by


def survey_8d_bytes(hb, sum_hits, mode):
    """SURVEY.md 8(d)'s ALG_PG / ALG_FRAME of the batch, kept for reference only (they price a full read of every
    bit-packed mask, which this design avoids): reported under `survey_8d`, never as a roofline fraction."""
    F, M = hb.n_frames, hb.n_masks
    W, H = hb.width, hb.height
    packed = M * ((W + 7) // 8) * H
    n_pts = hb.n_raw_rows
    pg = 16 * n_pts + packed + 4 * sum_hits + 4 * (M + F)
    masks_in = M * W * H if mode == "dense" else 4 * int(hb.rle_counts.size)
    return {"ALG_PG_per_frame": pg // F, "ALG_FRAME_per_frame": (pg + masks_in + packed + 16 * sum_hits + 64 * M) // F}


def visible_cores():
    """Cores this process may use: the scheduler affinity, cut by the cgroup CPU quota when there is one."""
    from cm3d_amd.reader import usable_cores
    return usable_cores()


def cpu_baseline(frames, lanes, frame_lane, hb, n_sample):
    """The CPU oracle (oracle/, a plain-C port of the reference algorithm in the reference's
    execution order: per-mask clone + re-projection, per-mask full-frame erode, O(M^2) medoid,
    f64 lane NN, circle NMS) timed on this box's host cores, single thread."""
    from oracle import oracle as orc
    from tests.helpers import oracle_batch
    orc.lib()
    n_sample = max(1, min(n_sample, len(frames)))
    sub = frames[:n_sample]
    sub_hb = lifting.pack_frames(sub, lanes, frame_lane[:n_sample])
    t0 = time.perf_counter()
    oracle_batch(orc, sub, lanes, frame_lane[:n_sample], sub_hb)
    dt = time.perf_counter() - t0
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n_sample / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"first {n_sample} frames of the same synthetic batch, {dt:.1f} s, host has {os.cpu_count()} cores ({cpu_model})"}


def cpu_baseline_all_cores(config_name, set_args, lane_points, workers, frames_per_worker):
    """The same oracle, one process per core over disjoint frames (frames are independent); aggregate rate =
    frames / slowest worker's oracle time.  SURVEY 8(d) asks for the all-cores figure beside the 1-thread one.
    The workers are plain child processes (oracle/cpu_worker.py); they never touch the GPU."""
    import subprocess
    procs = []
    for w in range(workers):
        cmd = [sys.executable, "-m", "oracle.cpu_worker", config_name, str(100000 + w * frames_per_worker), str(frames_per_worker),
               str(lane_points)] + list(set_args)
        procs.append(subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    res = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            pr.kill()
            raise
        if pr.returncode != 0:
            raise RuntimeError(f"cpu worker exited with {pr.returncode}")
        res.append(json.loads(out.strip().splitlines()[-1]))
    n = sum(r["frames"] for r in res)
    dt = max(r["seconds"] for r in res)
    return {"value": n / dt, "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"{workers} processes x {frames_per_worker} frames of the same synthetic workload, slowest {dt:.1f} s"}


def fusion_bench(n_samples, steps, warmup):
    """`--fusion N`: the SAM3D fusion matching (reference src/nuscenes/linear_matching.py:231-259) at
    nuScenes-val scale -- N samples, ~25 lifted boxes against ~45 SAM3D boxes each, one cm3d_bev_match call per step with
    the box records resident in HBM; the CPU oracle over the same samples beside it.  One JSON line."""
    import numpy as np
    from cm3d_amd import _lib, ops
    rng = np.random.default_rng(1)

    def boxes(n, centre, spread):
        c = centre + rng.uniform(-spread, spread, (n, 2))
        return np.stack([c[:, 0], c[:, 1], rng.uniform(-1, 1, n), rng.uniform(1.5, 5.5, n), rng.uniform(0.8, 2.5, n),
                         rng.uniform(1, 2, n), rng.uniform(-np.pi, np.pi, n)], 1)
    pr, gr = [], []
    for _ in range(n_samples):
        P = int(rng.integers(5, 45))
        p = boxes(P, rng.uniform(-1, 1, 2) * (700.0, 1500.0), 40.0)
        g = p.copy()
        g[:, :2] += rng.normal(0, 0.5, (P, 2)); g[:, 6] += rng.normal(0, 0.2, P)
        g = np.concatenate([g[rng.random(P) < 0.7], boxes(int(rng.integers(10, 50)), p[0, :2], 40.0)])
        pr.append(ops.match_records(p)); gr.append(ops.match_records(g))
    np_, ng = np.array([len(r) for r in pr]), np.array([len(r) for r in gr])
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    p_off, g_off = np.concatenate([[0], np.cumsum(np_)]).astype(np.int32), np.concatenate([[0], np.cumsum(ng)]).astype(np.int32)
    pair_off = np.concatenate([[0], np.cumsum(np_ * ng)]).astype(np.int64)
    d_p, d_g, d_po, d_go, d_pair = t(np.concatenate(pr)), t(np.concatenate(gr)), t(p_off), t(g_off), t(pair_off)
    n_pred, n_gt, total = int(p_off[-1]), int(g_off[-1]), int(pair_off[-1])
    pm, gm = torch.empty(n_pred, dtype=torch.int32, device=dev), torch.empty(n_gt, dtype=torch.int32, device=dev)
    iou = torch.empty(n_pred, dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = torch.empty(L.cm3d_bev_match_workspace_bytes(total), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(L.cm3d_bev_match(d_p.data_ptr(), d_po.data_ptr(), n_pred, d_g.data_ptr(), d_go.data_ptr(), n_gt, d_pair.data_ptr(),
                                    n_samples, total, 0.2, pm.data_ptr(), gm.data_ptr(), iou.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                    ws.numel(), st), "cm3d_bev_match")
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    from oracle import oracle as orc          # the checker, here as the reported CPU baseline
    t0 = time.perf_counter()
    exp = [orc.bev_match(a, b, 0.2)[0] for a, b in zip(pr, gr)]
    dt_cpu = time.perf_counter() - t0
    same = bool(np.array_equal(np.concatenate(exp), pm.cpu().numpy()))
    print(json.dumps({"metric": "SAM3D fusion matching samples/sec", "value": round(n_samples / dt, 1), "unit": "samples/s", "n_gpus": 1,
                      "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "dtype": "f64",
                      "data": "synthetic", "config": {"workload": f"{n_samples} samples, {n_pred} lifted x {n_gt} SAM3D boxes, {total} pairs"},
                      "matches": int((pm >= 0).sum().item()), "equals_oracle": same,
                      "cpu_baseline": {"value": round(n_samples / dt_cpu, 1), "unit": "samples/s", "cores": 1, "kind": "port",
                                       "sample": f"all {n_samples} samples, {dt_cpu:.2f} s"}}))


def self_launch(args, argv):
    """`--gpus N` without torchrun: start the N ranks as a child process (nothing in this process has touched the GPU:
    importing torch does not initialise HIP), relay its output -- rank 0's JSON line -- and return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def rehearse_launch(args):
    """`--rehearse-launch`: the launch + rendezvous + the single record exchange, and nothing else -- no kernels, so it
    also runs without a GPU (gloo).  Rank 0 prints one JSON line with the rank count the process group saw."""
    rank, world, local_rank = cdist.init_from_env()
    rec = torch.full((3 + rank, 10), float(rank), dtype=torch.float64)
    if world > 1 and torch.distributed.get_backend() == "nccl":
        rec = rec.to(f"cuda:{local_rank}")
    got = cdist.gather_records(rec, dst=0)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        ok = len(got) == world and all(g.shape[0] == 3 + r and bool((g == r).all()) for r, g in enumerate(got))
        print(json.dumps({"metric": "pseudo-label frames/sec on nuScenes-shaped sweeps", "value": None, "unit": "frames/s", "n_gpus": world,
                          "rehearsal": True, "ranks_in_group": torch.distributed.get_world_size() if world > 1 else 1,
                          "backend": torch.distributed.get_backend() if world > 1 else None, "gather_ok": bool(ok)}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def end_to_end_bench(args, quick=False):
    """quick=True: the leg of the DEFAULT bench line -- dataset of args.end_to_end frames, one warm-up run over two scenes, one timed run
    of the entry point with the native reader on threads; returns a dict instead of printing.
    `--end-to-end N`: what a user of the drop-in script pays -- files on disk -> pseudolabels_minival.json, through
    cm3d_amd.pipeline_nuscenes (src/nuscenes/2d_to_3d.py) on a synthetic nuScenes-layout dataset of N frames shaped like
    --config (C2 by default): table walk, <f>_data.json, mask pickles + RLE strings, sweep files, H2D, the GPU pass,
    the record gather, the JSON writer.  Timed three ways: the native loader (libcm3d_reader.so) on threads of this process,
    reader processes that each run the native loader, and -- on a slice -- the per-frame Python reader that mirrors the
    reference.  Inputs come from the page cache (the dataset was just written).  One JSON line."""
    import contextlib
    import io
    import shutil
    import tempfile
    from cm3d_amd import nusc_io, pipeline_nuscenes as pn
    cfg = syn.config(args.config)
    per_scene = 32
    n_scenes = max(2, (args.end_to_end + per_scene - 1) // per_scene)
    n_frames = n_scenes * per_scene
    root = tempfile.mkdtemp(prefix="cm3d_e2e_", dir=os.environ.get("CM3D_E2E_DIR") or None)
    try:
        t0 = time.perf_counter()
        dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(root, cfg, n_scenes=n_scenes, frames_per_scene=per_scene,
                                                                    lane_points=args.lane_points, pool=64)
        t_write = time.perf_counter() - t0
        sweep_bytes = sum(os.path.getsize(os.path.join(dataroot, "sweeps", "LIDAR_TOP", f)) for f in os.listdir(os.path.join(dataroot, "sweeps", "LIDAR_TOP")))
        cores = visible_cores()

        def run(tag, scenes, reader_threads, workers):
            out_dir = os.path.join(root, "out_" + tag)
            argv = ["--version", "v1.0-synth", "--dataroot", dataroot, "--mask-dir", mask_dir, "--output-dir", out_dir, "--scenes", ",".join(scenes),
                    "--ratio", str(cfg.ratio), "--n-sweeps", str(cfg.n_sweeps), "--scenes-per-batch", str(max(1, args.frames // per_scene)),
                    "--reader-threads", str(reader_threads), "--workers", str(workers), "--priors", os.path.join(ROOT, "src", "nuscenes", "cfg", "shape_priors_chatgpt.json")]
            buf = io.StringIO()
            t = time.perf_counter()
            with contextlib.redirect_stdout(buf):
                rc = pn.main(argv)
            dt = time.perf_counter() - t
            if rc != 0:
                raise RuntimeError(f"entry point returned {rc}")
            timer = {}
            for line in buf.getvalue().splitlines():
                if ":" in line and not line.startswith("wrote"):
                    k, v = line.split(":", 1)
                    try:
                        timer[k.strip()] = round(float(v), 3)
                    except ValueError:
                        pass
            return dt, json.load(open(os.path.join(out_dir, pn.OUTPUT_NAME))), timer

        run("warm", names[:2], 0, 0)                                      # first-use costs (library loads, allocator) stay out
        nproc = max(2, min(cores // 2, 16))
        dt_t, res_t, timer_t = run("threads", names, int(os.environ.get("CM3D_E2E_READER_THREADS", "0")), 0)
        if quick:
            return {"frames_per_s": round(n_frames / dt_t, 1), "seconds": round(dt_t, 3), "frames": n_frames, "timer": timer_t,
                    "boxes": sum(len(v) for v in res_t["results"].values()), "sweep_file_MB": round(sweep_bytes / 1e6, 1),
                    "usable_cores": cores, "frames_per_gpu_batch": args.frames, "dataset_write_seconds": round(t_write, 1),
                    "what": "files of a synthetic nuScenes-layout dataset (tables, .bin sweeps, _masks.pkl, _data.json) in the page cache -> "
                            "pseudolabels_minival.json through src/nuscenes/2d_to_3d.py's entry point (cm3d_amd.pipeline_nuscenes.main), "
                            "wall time of the call; the full comparison (reader processes, Python reader): bench.py --end-to-end N"}
        dt_p, res_p, timer_p = run("procs", names, max(1, cores // nproc), nproc)
        sub = names[:max(2, 128 // per_scene)]
        dt_py, res_py, _ = run("python", sub, -1, 0)
        same_procs = res_t == res_p
        same_python = all(res_t["results"][k] == v for k, v in res_py["results"].items())
        best = min(dt_t, dt_p)
        print(json.dumps({
            "metric": "pseudo-label frames/sec, files on disk -> pseudolabels_minival.json (entry point, end to end)",
            "value": round(n_frames / best, 1), "unit": "frames/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}-shaped nuScenes-layout dataset: {n_scenes} scenes x {per_scene} frames = {n_frames} frames "
                                   f"({cfg.n_points}x{cfg.n_sweeps} pts, {cfg.n_cams} cams, {cfg.n_masks} masks {cfg.width}x{cfg.height}), "
                                   f"{sweep_bytes / 1e6:.0f} MB of sweep files in the page cache, {args.frames} frames per GPU batch"},
            "native_reader_threads": {"frames_per_s": round(n_frames / dt_t, 1), "seconds": round(dt_t, 3), "threads": cores, "timer": timer_t},
            "native_reader_processes": {"frames_per_s": round(n_frames / dt_p, 1), "seconds": round(dt_p, 3), "processes": nproc,
                                        "threads_each": max(1, cores // nproc), "timer": timer_p},
            "python_reader": {"frames_per_s": round(len(sub) * per_scene / dt_py, 1), "frames": len(sub) * per_scene, "seconds": round(dt_py, 3),
                              "note": "pickle.load + np.fromfile per frame in one process, as the reference reads (2d_to_3d.py:422-441)"},
            "identical_output": {"threads_vs_processes": bool(same_procs), "native_vs_python_reader": bool(same_python)},
            "sweep_read_GBs": round(sweep_bytes / best / 1e9, 2), "usable_cores": cores, "dataset_write_seconds": round(t_write, 1),
            "samples": len(res_t["results"]), "boxes": sum(len(v) for v in res_t["results"].values())}))
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--config", default="c2", help="synthetic config (c1,c2,c4,c5,tiny)")
    ap.add_argument("--masks", default="rle", choices=["rle", "dense"],
                    help="mask input resident in HBM: COCO-RLE run lengths (the on-disk contract) or dense uint8")
    ap.add_argument("--cpu-sample", type=int, default=48, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--lane-points", type=int, default=50000)
    ap.add_argument("--no-secondary", action="store_true", help="skip the second mask mode")
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="processes of the all-cores CPU baseline (-1 = every core this job may use; 0 = skip)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a field of the synthetic config (e.g. --set n_masks=80); for experiments")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="independent batches kept in flight per GPU (LiftPipeline depth; 1 = one batch at a time).  r04: 4 with eight hardware "
                         "queues and the projection launch at one workgroup per CU: 2.46-2.48 M frames/s against 2.29 M with 3 (five and more: worse)")
    ap.add_argument("--keep-cloud", action="store_true", help="also materialise the transformed cloud (16 B/row more HBM traffic)")
    ap.add_argument("--reuse-batch", action="store_true",
                    help="generate ONE synthetic batch and make every slot in flight a resident copy of it (the large shapes: generating "
                         "three 256-frame batches of the 10-sweep configuration takes longer than the measurement)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("CM3D_BENCH_GRAPH", "0")),
                    help="1: the passes that carry no timing events are HIP-graph replays (one launch per pass instead of ~14)")
    ap.add_argument("--fusion", type=int, default=0, metavar="SAMPLES",
                    help="time the SAM3D fusion matching (SURVEY 8 f4) on this many samples instead of the lifting path")
    ap.add_argument("--rehearse-launch", action="store_true", help="launch, rendezvous and the record exchange only (no GPU needed)")
    ap.add_argument("--e2e-frames", type=int, default=int(os.environ.get("CM3D_BENCH_E2E_FRAMES", "2048")), metavar="FRAMES",
                    help="frames of the end-to-end leg of the default line (files -> labels through the entry point; 0 = skip)")
    ap.add_argument("--end-to-end", type=int, default=0, metavar="FRAMES",
                    help="time the entry point instead: files of a synthetic dataset of this many frames -> pseudolabels json")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args, argv)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        print(f"error: --gpus {args.gpus} but WORLD_SIZE={env_world}: start bench.py with --gpus equal to the number of ranks "
              "(or without torchrun: it launches the ranks itself)", file=sys.stderr)
        return 2
    if args.rehearse_launch:
        return rehearse_launch(args)
    if args.fusion > 0:
        fusion_bench(args.fusion, min(args.steps, 50), min(args.warmup, 5))
        return 0
    if args.end_to_end > 0:
        return end_to_end_bench(args)

    rank, world, local_rank = cdist.init_from_env()
    if os.environ.get("CM3D_SINGLE_DEVICE"):      # rehearsal of the N>1 path on a one-GPU box (with CM3D_DIST_BACKEND=gloo)
        local_rank = 0
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    over = {}
    for kv in args.set:
        k, v = kv.split("=", 1)
        over[k] = type(getattr(syn.SyntheticConfig(), k))(float(v)) if not isinstance(getattr(syn.SyntheticConfig(), k), str) else v
    cfg = syn.config(args.config, **over)
    depth = max(1, args.in_flight)
    t_gen = time.perf_counter()
    # one HD-map lane table covering the region all frames drive in (ego positions are drawn
    # from (600,1600) +- 200 m, objects up to 55 m further out)
    lanes = [syn.make_lane_table([600.0, 1600.0], args.lane_points, seed=7 + rank, extent=260.0)]
    frame_lane = [0] * args.frames
    # `depth` independent batches stay in flight (LiftPipeline: one engine + stream each); step k runs on batch k % depth
    batches = []
    cache_dir = os.environ.get("CM3D_BENCH_CACHE")     # experiments: packed batches kept between runs of one GPU call (generation takes
    for slot in range(depth):                          # longer than the measurement); only without the CPU legs, which need the frames
        import pickle
        # (rank 0 needs the frames themselves for the CPU legs; every other rank only the packed batch)
        cache = cache_dir and (args.cpu_sample == 0 or rank != 0) and os.path.join(
            cache_dir, f"bench_{args.config}_{args.frames}_{rank * depth + slot}_{args.lane_points}_{'_'.join(sorted(args.set))}.pkl")
        if cache and os.path.exists(cache):
            batches.append((None, pickle.load(open(cache, "rb"))))
            continue
        if args.reuse_batch and batches:
            batches.append(batches[0])
            continue
        fr = []
        for i in range(args.frames):
            fr.append(syn.make_frame(cfg, (rank * depth + slot) * args.frames + i))
            if rank == 0 and cfg.n_points * cfg.n_sweeps >= 300000 and (i + 1) % 32 == 0:
                print(f"generating: batch {slot}, frame {i + 1} / {args.frames}", file=sys.stderr, flush=True)     # a long run must not look hung
        batches.append((fr, lifting.pack_frames(fr, lanes, frame_lane)))
        if cache:
            os.makedirs(cache_dir, exist_ok=True)
            pickle.dump(batches[-1][1], open(cache, "wb"), protocol=4)
    frames, hb = batches[0]
    t_gen = time.perf_counter() - t_gen

    modes = [args.masks] + ([] if args.no_secondary else [m for m in ("rle", "dense") if m != args.masks])
    pipe = lifting.LiftPipeline(dev, depth=depth, keep_cloud=args.keep_cloud)
    for slot in range(depth):
        with torch.cuda.stream(pipe.streams[slot]):
            pipe.engines[slot].upload(batches[slot][1])
            if "dense" in modes:
                pipe.engines[slot].decode_masks_dense()
    eng = pipe.engines[0]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    results = {}
    for mode in modes:
        fused = eng.can_fuse_sweeps()
        stages = [s for s in eng.STAGES if not (fused and s == "sweeps")]
        ev = {s: [] for s in stages}

        def lanes_after_grid(s):
            eng.wait_lane_grid()        # the lane-grid build runs on the engine's side stream since stage_begin
            eng.stage_lanes(s)

        calls = {"sweeps": eng.stage_sweeps, "masks": lambda s: eng.stage_masks(s, mode),
                 "project": eng.stage_sweep_project if fused else eng.stage_project,
                 "compact": eng.stage_compact, "medoid": eng.stage_medoid, "lanes": lanes_after_grid, "boxes": eng.stage_boxes}
        # Everything that is not a pass happens BEFORE the warm-up steps, so that the timed region follows them with nothing but the barrier
        # and the synchronize in between: a few milliseconds of host work there (status read-backs, event creation) let the GPU's clocks
        # drop, and the first region measured 3 % (three batches in flight) to 17 % (four) below every later one.
        if world > 1 and mode == modes[0]:
            cdist.gather_records(eng.b.box, dst=0)      # untimed: sets up the RCCL channels the final gather uses (the records' buffer exists since upload)
        ev_every = 1 if args.steps <= 16 else 8        # an event pair costs ~5 us of stream time: sample every 8th step
        for _ in range(0, args.steps, ev_every):        # created (first record) outside the timed region
            pe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pe[0].record()
            pe[1].record()
            ev["project"].append(pe)
        graphs = None
        if args.graph:
            for k in range(depth):
                pipe.rerun(k, masks=mode)               # (a capture wants the once-only work of a slot's first pass done)
            torch.cuda.synchronize()
            graphs = []
            for slot in range(depth):
                with torch.cuda.stream(pipe.streams[slot]):
                    graphs.append(pipe.engines[slot].capture_graph(masks=mode))
            torch.cuda.synchronize()
        for k in range(max(args.warmup, depth)):
            pipe.rerun(k % depth, masks=mode)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(args.steps):
            pe = None
            if step % ev_every == 0:    # HIP events around the roofline kernel only, recorded by the library on its launch stream
                pe = ev["project"][step // ev_every]
            if graphs is not None and pe is None:
                with torch.cuda.stream(pipe.streams[step % depth]):
                    graphs[step % depth].replay()
            else:
                pipe.rerun(step % depth, masks=mode, project_events=pe)       # LiftEngine.run() on that batch's stream
        gathered = None
        if mode == modes[0]:
            # the single end-of-job exchange: fixed-size box records -> rank 0 (RCCL gather).  With more than one rank the records must be
            # final first: a host synchronize (cross-stream event waits on the current stream -- wait_stream -- cost 0.4 ms of a 34 ms region
            # with three batches in flight and 3.7 ms with four); one rank keeps its records, nothing to wait for before the synchronize below.
            if world > 1:
                torch.cuda.synchronize()
            gathered = cdist.gather_records(eng.b.box, dst=0)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for e in pipe.engines:
            e.check_status()                            # (of the warm-up and the timed passes; raises on any error flag)
        per_rank_dt = [dt]
        if world > 1:
            mine = torch.tensor([dt], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(every, mine)
            per_rank_dt = [float(t.item()) for t in every]
            dt = max(per_rank_dt)                        # the job's time is its slowest rank's
        # how far a K-step region can be off: the same K passes five more times, and one region of >= 0.5 s
        spread, long_rate = None, None
        if mode == modes[0]:
            def region(n_steps):
                barrier()
                torch.cuda.synchronize()
                t_r = time.perf_counter()
                for step in range(n_steps):
                    pipe.rerun(step % depth, masks=mode)
                region.enqueue_s = time.perf_counter() - t_r        # the host is done handing the passes to the streams here
                torch.cuda.synchronize()
                barrier()
                torch.cuda.synchronize()
                d = time.perf_counter() - t_r
                if world > 1:
                    mine_r = torch.tensor([d], dtype=torch.float64, device=dev)
                    every_r = [torch.zeros_like(mine_r) for _ in range(world)]
                    torch.distributed.all_gather(every_r, mine_r)
                    d = max(float(t.item()) for t in every_r)
                return d
            reps = [args.frames * world * args.steps / region(args.steps) for _ in range(5)]
            spread = {"min": round(min(reps), 1), "median": round(float(np.median(reps)), 1), "max": round(max(reps), 1), "repeats": 5}
            n_long = int(max(args.steps, min(20000, np.ceil(0.5 / max(dt / args.steps, 1e-6)))))
            long_rate = {"value": round(args.frames * world * n_long / region(n_long), 1), "steps": n_long}
            long_rate["host_enqueue_ms_per_step"] = round(region.enqueue_s / n_long * 1e3, 4)
            # what the timed loop amortises and a job of ever-new batches does not: the lane index is built once for the
            # resident tables, and the medoid stage runs on the hint of the pass before.  The same K passes with the index rebuilt
            # in EVERY pass (each slot on an index of its own: a rebuild must not touch what another slot's pass is reading) and
            # without the hint:
            for e in pipe.engines:
                ln = dict(e._lane)
                ln["grid"] = e._lane["grid"].clone()
                ln["built_event"] = torch.cuda.Event()
                e._lane, e.b.grid = ln, ln["grid"]
            hints = [e._md_hint for e in pipe.engines]

            def region_unamortised(n_steps):
                barrier()
                torch.cuda.synchronize()
                t_r = time.perf_counter()
                for step in range(n_steps):
                    e = pipe.engines[step % depth]
                    e.rebuild_lane_grid()
                    e._md_hint = False
                    pipe.rerun(step % depth, masks=mode)
                torch.cuda.synchronize()
                barrier()
                torch.cuda.synchronize()
                return time.perf_counter() - t_r
            region_unamortised(depth)
            unamortised = round(args.frames * world * args.steps / region_unamortised(args.steps), 1)
            for e, h in zip(pipe.engines, hints):
                e._md_hint = h
        # per-stage breakdown: a separate, untimed pass over ONE batch alone, with events around every stage -- with the projection launch
        # as one batch alone gets it (the whole chip; the pipeline above asked for two workgroups per CU: cm3d_project_workgroups_per_cu)
        wg_hint = eng.lib.cm3d_project_workgroups_per_cu(0)
        alone = []
        with torch.cuda.stream(pipe.streams[0]):
            st = pipe.streams[0].cuda_stream
            for _ in range(max(3, min(args.steps, 10))):
                eng.stage_begin(st)
                for s in stages:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    if s == "project":      # the library records the pair around the projection kernel itself ...
                        b.record()
                        c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        c.record()          # ... and this pair, like every other stage's, sits outside the call: it also holds the
                        calls[s](st, (a, b))    # per-frame table kernel the call launches first and the launch boundaries
                        d.record()
                        alone.append((a, b))
                        ev.setdefault("project_with_tables", []).append((c, d))
                    else:
                        calls[s](st)
                        b.record()
                        ev[s].append((a, b))
        torch.cuda.synchronize()
        eng.lib.cm3d_project_workgroups_per_cu(wg_hint)
        # (project: the events of the timed region -> mean, it is an average launch duration; the alone-pass stages: medians of a
        # handful of passes, one preempted pass must not stand for the stage)
        stage_ms = {s: float(np.mean([a.elapsed_time(b) for a, b in ev[s]])) if s == "project" else float(np.median([a.elapsed_time(b) for a, b in ev[s]]))
                    for s in list(stages) + ["project_with_tables"]}
        project_alone_ms = float(np.median([a.elapsed_time(b) for a, b in alone]))
        status = eng.check_status()
        results[mode] = dict(dt=dt, per_rank_dt=per_rank_dt, stage_ms=stage_ms, project_alone_ms=project_alone_ms, fused=fused, sum_hits=int(status[2]),
                             hit_rows=eng.hit_chunk_rows(),
                             n_points=int(status[1]), sum_pairs=int((eng.b.hit_count.to(torch.int64) ** 2).sum().item()),
                             sum_pairs_long=int((eng.b.hit_count.to(torch.int64) ** 2)[(eng.b.hit_count > 256) & (eng.b.hit_count < 100000)].sum().item())
                             if int(eng.b.hit_count.max().item()) > 448 else 0,      # (csrc/medoid.hip: MD_LONG_MIN in a batch with a list beyond MD_BATCH_LONG)
                             n_boxes=int((eng.b.flags == 3).sum().item()), max_hits=int(eng.b.hit_count.max().item()),
                             rect_bytes=packed_rect_bytes(eng.b.bbox.cpu().numpy(), eng.b.Wp),
                             n_gathered=None if gathered is None else int(sum(g.shape[0] for g in gathered)), spread=spread, long_rate=long_rate, unamortised=unamortised if mode == modes[0] else None)

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return 0
    main_mode = modes[0]
    r = results[main_mode]
    frames_total = args.frames * world * args.steps
    value = frames_total / r["dt"]
    cloud_stored = eng.b.points is not None
    by = compulsory_bytes(hb, eng.b.planes, r["sum_hits"], main_mode, r["fused"], cloud_stored, r["rect_bytes"], r["hit_rows"])

    def rate(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9

    def roof(kernel_key, ms, kernel_name, note):
        ach = rate(by[kernel_key], ms)
        return {"kernel": kernel_name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": int(by[kernel_key]),
                "avg_launch_ms": round(ms, 4), "note": note}

    # north_star's kernel: projection + in-mask test (one launch per pass), timed by HIP events on its launch stream
    # inside the timed region -- i.e. while the other batches in flight share the GPU with it
    rows = hb.n_raw_rows
    per_row_in = (hb.bytes_per_row if r["fused"] else 16) + (16 if (r["fused"] and cloud_stored) else 0)
    roofline = roof("k_project_hits", r["stage_ms"]["project"], ("k_project_q<NPL, KEEP> (quad layout)" if hb.quads else "k_project_hits<ONE_PLANE, FUSED, STRIDE>") if r["fused"] else "k_project_hits",
                    f"bytes that must cross HBM per launch = {per_row_in} B/row x {rows} rows: "
                    + (f"raw sweep rows read ({hb.bytes_per_row} B/row, layout {'quads: x,y,z of four rows side by side, the unused columns stay on the host' if hb.quads else 'rows as in the .bin files'})" if r["fused"] else "prepared cloud read (16 B/row)")
                    + (" + transformed cloud written (16 B/row)" if (r["fused"] and cloud_stored) else "")
                    + f" + hit words written, 4 B/row/plane ({eng.b.planes} plane(s)) for the {r['hit_rows']} rows in 256-row blocks that hold an "
                      "in-mask point (the others' words are neither written nor read since r03; cm3d_project_hit_rows counts them on the device's flags)"
                    + "; the bit-packed mask words the launch gathers (bounding-box gated, mostly L2 hits) and the per-frame tables "
                      "are extra traffic and NOT counted, so `frac` cannot be inflated by bytes the kernel skips; avg_launch_ms = HIP events the "
                      "library records on the launch stream right around the kernel, inside the timed region: with several batches in flight "
                      "it holds the time the launch queues behind and shares the chip with the other batches' kernels (rocprof's execution time "
                      "of the same launches: profiles/*_kernel_stats.csv) -- and since r04 the pipeline GIVES the launch only a part of the chip while batches are "
                      "in flight (workgroups_per_cu_in_flight: one workgroup per CU of the three that fit with four batches in flight; the pass is faster "
                      "for it, this launch slower), so `frac` says what the launch gets of the HBM while sharing, not what the kernel can do; "
                      "frac_alone = the same launch with nothing else on the GPU and the whole chip, the figure to judge the kernel by")
    per_row = round(by["k_project_hits"] / max(1, rows), 2)
    roofline["rows_per_launch"] = rows
    roofline["bytes_per_row"] = per_row
    # the box's own streaming rate beside the nominal peak: device-to-device copy of 1 GiB, read + write bytes
    src_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst_buf = torch.empty_like(src_buf)
    dst_buf.copy_(src_buf)
    torch.cuda.synchronize()
    t_cp = time.perf_counter()
    for _ in range(5):
        dst_buf.copy_(src_buf)
    torch.cuda.synchronize()
    roofline["measured_copy_GBs"] = round(5 * 2 * src_buf.numel() * 4 / (time.perf_counter() - t_cp) / 1e9, 1)
    del src_buf, dst_buf
    roofline["batches_in_flight"] = depth
    roofline["workgroups_per_cu_in_flight"] = wg_hint if wg_hint else "as many as fit minus one (3)"      # cm3d_project_workgroups_per_cu; the alone figures: the whole chip
    roofline["avg_launch_ms_alone"] = round(r["project_alone_ms"], 4)      # the same launch with nothing else on the GPU
    roofline["frac_alone"] = round(rate(by["k_project_hits"], r["project_alone_ms"]) / HBM_PEAK_GBS, 4)
    # the same against what a plain stream gets on THIS box (the device copy above: reads + writes), not the nominal peak
    if roofline["measured_copy_GBs"] > 0:
        roofline["frac_alone_vs_measured_copy"] = round(rate(by["k_project_hits"], r["project_alone_ms"]) / roofline["measured_copy_GBs"], 4)
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    per_kernel = {}
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            # the PMC passes were taken on the default workload; a different batch gets no traffic figure
            if args.frames == int(tr.get("frames_per_gpu", 256)) and not args.set and not args.keep_cloud:
                per_kernel = tr.get(f"{args.config}_{main_mode}", {})
                roofline["traffic"] = per_kernel.get("k_project_q" if hb.quads else "k_project_hits")
                # The launch is bound by the vector pipe, not by HBM: SQ_INSTS_VALU of the launch (its own PMC pass,
                # profiles/traffic.json) x 4 cycles -- what SQ_ACTIVE_INST_VALU charges a vector instruction of this kernel's mix, and what
                # tools/ubench/valu_wall.hip measures for its packed / compare / min / max / DPP instructions -- over the cycles 1024 SIMDs
                # have in the launch's time alone at the nominal 2.4 GHz (the clock under load is lower: the true share is higher)
                pv = per_kernel.get(("k_project_q" if hb.quads else "k_project_hits") + "_insts_valu")
                if pv:
                    roofline["valu_pipe"] = {"vector_instructions_per_launch": int(pv), "cycles_per_instruction": 4.0,
                                             "frac_alone_at_2.4GHz": round(pv * 4.0 / (1024 * r["project_alone_ms"] * 1e-3 * 2.4e9), 4),
                                             "note": "share of the SIMDs' vector-pipe cycles the launch's instructions occupy while it runs alone"}
                if roofline["traffic"] and roofline.get("measured_copy_GBs", 0) > 0:      # measured HBM bytes of the launch over its time alone
                    roofline["traffic_frac_alone_vs_measured_copy"] = round(rate(roofline["traffic"], r["project_alone_ms"]) / roofline["measured_copy_GBs"], 4)
        except (OSError, ValueError):
            pass
    mask_kernel = "k_erode_pack" if main_mode == "dense" else "k_rle_erode_pack_wave"
    kernels = {
        "masks": roof(mask_kernel, r["stage_ms"]["masks"], mask_kernel,
                      "run lengths read + the words of every eroded mask's bounding rectangle written (what the kernel stores); "
                      "instruction-issue and latency bound, one wave per mask; stage time incl. the launch boundary" if main_mode == "rle" else
                      "dense uint8 masks read + bit-packed masks written; stage time incl. the launch boundary"),
        "stage_ms_one_batch_alone": {k: round(v, 4) for k, v in dict(r["stage_ms"], project=r["project_alone_ms"]).items()},
    }
    kernels["masks"]["traffic"] = per_kernel.get(mask_kernel)
    # north_star's "projection + gather" as a GROUP: the projection launch and the compaction behind it (k_compact_hits: ranks
    # and index lists; k_hit_xyz: the listed points' coordinates), bytes and times summed, one batch alone on the GPU
    grp_ms = r["project_alone_ms"] + r["stage_ms"]["compact"]
    grp_bytes = by["k_project_hits"] + by["k_compact_hits"]
    roofline_group = {"kernels": ["k_project_hits", "k_compact_hits", "k_hit_xyz"], "bound": "hbm", "bytes": int(grp_bytes),
                      "ms_alone": round(grp_ms, 4), "achieved": round(rate(grp_bytes, grp_ms), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(rate(grp_bytes, grp_ms) / HBM_PEAK_GBS, 4),
                      "note": "projection kernel (library events) + the compaction stage (events outside the call: two launches and "
                              "their boundaries), one batch alone; compaction bytes = hit words read + 4 B index + 16 B coordinates "
                              "written per listed point" + ("" if cloud_stored else " + 12 B of its raw row read")}
    # The medoid against the VECTOR-ALU roof ('report its time separately, not against HBM').  pairs = sum over masks of
    # (points in the mask)^2.  Priced in vector-instruction ISSUE SLOTS, the unit the kernel is bound by (MI355X_MICROARCH.md,
    # 'vector-instruction ISSUE cost': 4 cycles per wave instruction, packed float32 included, 8 for v_rsq / v_sqrt): a SIMD issues
    # 2.4e9 / 4 slots a second, 1024 SIMDs 614 G, each slot serves 64 lanes.  ALGORITHMIC slots per pair, i.e. what the
    # reference's float32 arithmetic costs in this instruction set with nothing wasted:
    #   exact route (lists up to 256 points, or of a batch without a list beyond 448: csrc/medoid.hip md_rows): 5 packed operations
    #     per two pairs for the cdist expansion (2.5), clamp_min_ (1), v_rsq (2), its cap for zeros (1), 7 packed operations per two
    #     pairs for the correctly rounded root (3.5), the ordered sum (1)                                           = 11 slots
    #   first pass of the long lists (md_approx_tile): five v_mfma_f32_32x32x1_2b_f32 per 2048 pairs, 64 cycles each ON THE SAME PIPE --
    #     float32 matrix instructions and vector instructions do not overlap on this chip (tools/ubench/mfma_sqrt.hip,
    #     profiles/r04_mfma_sqrt.txt) -- (2.5), v_sqrt with the output clamp (2), the packed sum (0.5)              = 5 slots
    #     (r03 priced 3.5: it took the matrix pipe for free and charged the clamp)
    # frac = slots the pairs need / slots the stage's time offers; issued_slots_per_pair (PMC SQ_INSTS_VALU of the same launch,
    # idle lanes of partly filled tiles, staging, tails and the second pass of the long lists included) says how far the
    # kernel's own instruction stream is from the algorithmic one.
    VALU_ISSUE_SLOTS = 256 * 4 * 2.4e9 / 4.0
    md_ms = r["stage_ms"]["medoid"]
    insts = per_kernel.get("k_medoid_insts_valu")
    pairs_long = r.get("sum_pairs_long", 0)
    alg_slots = 11.0 * (r["sum_pairs"] - pairs_long) + 5.0 * pairs_long
    offered = VALU_ISSUE_SLOTS * 64.0 * md_ms * 1e-3
    kernels["medoid"] = {"kernel": "k_medoid_tiles + k_medoid_reduce + k_medoid_long", "bound": "valu", "pairs_per_launch": r["sum_pairs"],
                         "pairs_in_long_lists": pairs_long, "stage_ms_alone": round(md_ms, 4),
                         "achieved_Gpairs_s": round(r["sum_pairs"] / (md_ms * 1e-3) / 1e9, 1),
                         "algorithmic_slots_per_pair": round(alg_slots / max(1, r["sum_pairs"]), 2),
                         "issued_slots_per_pair": None if not (insts and r["sum_pairs"]) else round(insts * 64.0 / r["sum_pairs"], 1),
                         "peak_Gslots_s": round(VALU_ISSUE_SLOTS / 1e9, 1), "unit": "wave-instruction issue slots/s",
                         "frac": round(alg_slots / offered, 4) if offered else None,
                         "note": "pairs = sum of M^2 over the masks' index lists; the float32 arithmetic per pair is the reference's "
                                 "(torch.cdist expansion + correctly rounded sqrt + ordered sum); stage time incl. three launch boundaries; "
                                 "issued_slots_per_pair is measured on the headline shape (profiles/traffic.json) and absent elsewhere"}
    if not r["fused"]:
        kernels["sweeps"] = roof("k_sweep_xform", r["stage_ms"]["sweeps"], "k_sweep_xform", "HBM streaming, one pass; stage time incl. the launch boundary")
        kernels["sweeps"]["traffic"] = per_kernel.get("k_sweep_xform")
    pass_rate = by["pass_total"] * (value / world / args.frames) / 1e9       # bytes of one pass x passes per second per GPU
    out = {
        "metric": "pseudo-label frames/sec on nuScenes-shaped sweeps", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r["dt"] / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {args.frames} frames/GPU x ({cfg.n_points}x{cfg.n_sweeps} pts, {cfg.n_cams} cams, "
                               f"{cfg.n_masks} masks {cfg.width}x{cfg.height}), masks resident as {main_mode}",
                   "frames_per_gpu": args.frames, "points_per_frame": cfg.n_points * cfg.n_sweeps, "masks_per_frame": cfg.n_masks,
                   "mask_size": [cfg.width, cfg.height], "lane_points": args.lane_points, "mask_input": main_mode,
                   "batches_in_flight": depth, "distinct_batches": 1 if args.reuse_batch else depth, "cloud_materialised": cloud_stored,
                   "parallelism": f"frame-sharded x{world}, {depth} independent batches in flight per GPU, one RCCL gather of box records"},
        # the timed K steps are short (a few ms at the driver's --steps): the same region five more times, and one of >= 0.5 s
        "value_spread": r["spread"], "value_long": None if r["long_rate"] is None else r["long_rate"]["value"],
        # what the HOST needs to hand one pass to its stream (Python + ~12 launches), measured in the long region: the pass is GPU-bound while
        # this stays below ms_per_step
        "host_enqueue_ms_per_step": None if r["long_rate"] is None else r["long_rate"].get("host_enqueue_ms_per_step"),
        "value_long_steps": None if r["long_rate"] is None else r["long_rate"]["steps"],
        # the same K passes with the lane index rebuilt in every pass and the medoid stage without the previous pass's hint: what a
        # job pays whose every batch brings new lane tables (the timed loop replays resident batches: index built once, hint warm)
        "value_unamortised": r["unamortised"],
        "ranks_in_group": torch.distributed.get_world_size() if world > 1 else 1,
        "per_rank_ms_per_step": [round(t / args.steps * 1e3, 4) for t in r["per_rank_dt"]],
        "roofline": roofline,
        "roofline_group": roofline_group,
        "kernels": kernels,
        # the whole pass against the HBM roof: bytes every stage must move (compulsory_bytes) x passes/s, per GPU
        "pass_roofline": {"bound": "hbm", "achieved": round(pass_rate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(pass_rate / HBM_PEAK_GBS, 4), "bytes_per_pass": int(by["pass_total"]),
                          "note": "per GPU; a pass is a chain of dependent launches, most of them latency- or VALU-bound (medoid, "
                                  "lane search): the fraction says how far the WHOLE path is from streaming its compulsory bytes"},
        "survey_8d": dict(survey_8d_bytes(hb, r["sum_hits"], main_mode),
                          note="SURVEY 8(d)'s per-frame figures, which price a full read of every bit-packed mask; this design never "
                               "moves those bytes, so they are reference numbers only and enter no fraction"),
        "boxes_per_step": r["n_boxes"], "in_mask_points_per_step": r["sum_hits"], "max_points_in_a_mask": r["max_hits"],
        "gen_seconds": round(t_gen, 1),
    }
    for mode in modes[1:]:
        o = results[mode]
        out[f"value_{mode}_masks"] = round(frames_total / o["dt"], 1)
        out[f"stage_ms_{mode}_masks"] = {k: round(v, 4) for k, v in dict(o["stage_ms"], project=o["project_alone_ms"]).items()}
    if world == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(frames, lanes, frame_lane, hb, args.cpu_sample)
        cores = visible_cores()
        workers = cores if args.cpu_workers < 0 else min(args.cpu_workers, cores)
        # every worker imports torch + numpy (~0.4 GB resident): stay far below the box's host-memory cap
        workers = min(workers, 192)
        if workers > 1:
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.config, args.set, args.lane_points, workers,
                                                                       max(4, min(16, args.cpu_sample // 4)))
                out["cpu_baseline_all_cores"]["sample"] += f"; host has {os.cpu_count()} cores, {cores} usable by this job (affinity / cgroup quota)"
            except Exception as exc:        # a reported extra: never let it break the bench line
                out["cpu_baseline_all_cores"] = {"error": repr(exc)}
    else:
        out["cpu_baseline"] = None
    # what a user of the drop-in script gets: files on disk -> labels, timed here so that the driver's run carries it
    if world == 1 and args.e2e_frames > 0 and args.cpu_sample > 0:
        try:
            del pipe, eng, batches
            torch.cuda.empty_cache()
            import copy
            e_args = copy.copy(args)
            e_args.end_to_end = args.e2e_frames
            out["end_to_end"] = end_to_end_bench(e_args, quick=True)
        except Exception as exc:            # a reported extra: never let it break the bench line
            out["end_to_end"] = {"error": repr(exc)}
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except cdist.GatherError as exc:        # a rank missing / a timeout in the one exchange: say so and leave with a code of its own
        print(f"error: {exc}", file=sys.stderr, flush=True)
        os._exit(3)                         # (not sys.exit: a half-dead process group can hang interpreter shutdown)
