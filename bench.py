#!/usr/bin/env python3
"""Benchmark of the lifting hot path (BASELINE.json: pseudo-label frames/sec on
nuScenes-shaped sweeps).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

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

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cm3d_amd import dist as cdist, lifting, synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(hb, sum_hits, mode):
    """SURVEY.md 8(d).  Returns per-kernel algorithmic bytes of ONE pass over the batch."""
    F, M = hb.n_frames, hb.n_masks
    W, H = hb.width, hb.height
    packed = M * ((W + 31) // 32) * 4 * H
    n_raw = hb.n_raw_rows
    n_pts = n_raw      # upper bound; the ego-box filter removes ~0.2 %
    alg = {
        # ALG_PG = 16 N + n*ceil(W*H/8) + 4*sum(M_i) + 4*(n+1), summed over the frames of the batch
        "project_gather": 16 * n_pts + packed + 4 * sum_hits + 4 * (M + F),
        "k_project_hits": 16 * n_pts + packed + 4 * n_pts,
        "k_erode_pack": M * W * H + packed,
        "k_rle_erode_pack": 4 * int(hb.rle_counts.size) + packed,
        "k_sweep": 4 * hb.raw_stride * n_raw + 16 * n_pts,
    }
    # sweep preparation folded into the projection launch: raw rows read, cloud written once, masks, hit words
    alg["k_project_hits_fused"] = alg["k_sweep"] + packed + 4 * n_pts
    masks_in = M * W * H if mode == "dense" else 4 * int(hb.rle_counts.size)
    alg["frame_total"] = alg["project_gather"] + masks_in + packed + 16 * sum_hits + 64 * M
    return alg


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


def main():
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
    ap.add_argument("--cpu-workers", type=int, default=16,
                    help="processes of the all-cores CPU baseline (0 = skip; capped at the visible cores)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a field of the synthetic config (e.g. --set n_masks=80); for experiments")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="independent batches kept in flight per GPU (LiftPipeline depth; 1 = one batch at a time)")
    ap.add_argument("--fusion", type=int, default=0, metavar="SAMPLES",
                    help="time the SAM3D fusion matching (SURVEY 8 f4) on this many samples instead of the lifting path")
    args = ap.parse_args()
    if args.fusion > 0:
        return fusion_bench(args.fusion, min(args.steps, 50), min(args.warmup, 5))

    rank, world, local_rank = cdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
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
    for slot in range(depth):
        fr = [syn.make_frame(cfg, (rank * depth + slot) * args.frames + i) for i in range(args.frames)]
        batches.append((fr, lifting.pack_frames(fr, lanes, frame_lane)))
    frames, hb = batches[0]
    t_gen = time.perf_counter() - t_gen

    modes = [args.masks] + ([] if args.no_secondary else [m for m in ("rle", "dense") if m != args.masks])
    pipe = lifting.LiftPipeline(dev, depth=depth)
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
        for k in range(max(args.warmup, depth)):
            pipe.rerun(k % depth, masks=mode)
        torch.cuda.synchronize()
        if world > 1 and mode == modes[0]:
            cdist.gather_records(eng.b.box, dst=0)      # untimed: sets up the RCCL channels the final gather uses
        torch.cuda.synchronize()
        for e in pipe.engines:
            e.check_status()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_every = 1 if args.steps <= 16 else 8        # an event pair costs ~5 us of stream time: sample every 8th step
        for step in range(args.steps):
            pe = None
            if step % ev_every == 0:    # HIP events around the roofline kernel only, on its launch stream
                pe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev["project"].append(pe)
            pipe.rerun(step % depth, masks=mode, project_events=pe)       # LiftEngine.run() on that batch's stream
        gathered = None
        if mode == modes[0]:
            # the single end-of-job exchange: fixed-size box records -> rank 0 (RCCL gather)
            for s in pipe.streams:
                torch.cuda.current_stream(dev).wait_stream(s)
            gathered = cdist.gather_records(eng.b.box, dst=0)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        # per-stage breakdown: a separate, untimed pass over ONE batch alone, with events around every stage
        alone = []
        with torch.cuda.stream(pipe.streams[0]):
            st = pipe.streams[0].cuda_stream
            for _ in range(max(3, min(args.steps, 10))):
                eng.stage_begin(st)
                for s in stages:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    calls[s](st)
                    b.record()
                    (alone if s == "project" else ev[s]).append((a, b))
        torch.cuda.synchronize()
        stage_ms = {s: float(np.mean([a.elapsed_time(b) for a, b in ev[s]])) for s in stages}
        project_alone_ms = float(np.mean([a.elapsed_time(b) for a, b in alone]))
        status = eng.check_status()
        results[mode] = dict(dt=dt, stage_ms=stage_ms, project_alone_ms=project_alone_ms, fused=fused, sum_hits=int(status[2]),
                             n_points=int(status[1]),
                             n_boxes=int((eng.b.flags == 3).sum().item()), max_hits=int(eng.b.hit_count.max().item()),
                             n_gathered=None if gathered is None else int(sum(g.shape[0] for g in gathered)))

    if rank != 0:
        return
    main_mode = modes[0]
    r = results[main_mode]
    frames_total = args.frames * world * args.steps
    value = frames_total / r["dt"]
    alg = algorithmic_bytes(hb, r["sum_hits"], main_mode)

    def roof(kernel_key, stage_key, res, kernel_name, note):
        ms = res["stage_ms"][stage_key]
        ach = alg[kernel_key] / (ms * 1e-3) / 1e9
        return {"kernel": kernel_name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "alg_bytes_per_launch": int(alg[kernel_key]),
                "avg_launch_ms": round(ms, 4), "note": note}

    # north_star's kernel: projection + in-mask gather (one launch per pass), timed by HIP events on its launch stream
    # inside the timed region -- i.e. while the other batches in flight share the GPU with it
    if r["fused"]:
        roofline = roof("k_project_hits_fused", "project", r, "k_project_hits<ONE_PLANE, FUSED>",
                        "sweep preparation folded in: algorithmic bytes = raw rows read (4 x stride B) + cloud written (16 B/point) + "
                        "every bit-packed mask once + 4 B/point hit word (SURVEY 8d); the bounding-box test lets the kernel skip most "
                        "mask bytes, so `achieved` can exceed what HBM allows for a full read; traffic_rate = measured HBM bytes / "
                        "time; the kernel is instruction-issue / latency bound")
    else:
        roofline = roof("k_project_hits", "project", r, "k_project_hits",
                        "algorithmic bytes = 16 B/point + every bit-packed mask once + 4 B/point hit word (SURVEY 8d); the kernel's "
                        "bounding-box test lets it skip most mask bytes, so `achieved` exceeds what HBM allows for a full read; "
                        "traffic_rate = measured HBM bytes / time; the kernel is instruction-issue bound")
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
    roofline["avg_launch_ms_alone"] = round(r["project_alone_ms"], 4)      # the same launch with nothing else on the GPU
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    per_kernel = {}
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            # the PMC passes were taken on the default workload; a different batch gets no traffic figure
            if args.frames == int(tr.get("frames_per_gpu", 256)) and not args.set:
                per_kernel = tr.get(f"{args.config}_{main_mode}", {})
                roofline["traffic"] = per_kernel.get("k_project_hits")
            if roofline["traffic"]:
                # what HBM actually moved per launch / time: the honest distance from the HBM roof (the kernel is
                # bound by instruction issue; `frac` above is SURVEY 8(d)'s algorithmic figure)
                rate = roofline["traffic"] / (roofline["avg_launch_ms"] * 1e-3) / 1e9
                roofline["traffic_rate"] = round(rate, 1)
                roofline["traffic_frac"] = round(rate / HBM_PEAK_GBS, 4)
        except (OSError, ValueError):
            pass
    mask_kernel = "k_erode_pack" if main_mode == "dense" else "k_rle_erode_pack"
    kernels = {
        "masks": roof(mask_kernel, "masks", r, mask_kernel, "latency-bound (one workgroup per mask); stage time incl. the launch boundary"),
        "stage_ms_one_batch_alone": {k: round(v, 4) for k, v in dict(r["stage_ms"], project=r["project_alone_ms"]).items()},
    }
    kernels["masks"]["traffic"] = per_kernel.get(mask_kernel)
    if not r["fused"]:
        kernels["sweeps"] = roof("k_sweep", "sweeps", r, "k_sweep_xform", "HBM streaming, one pass; stage time incl. the launch boundary")
        kernels["sweeps"]["traffic"] = per_kernel.get("k_sweep_xform")
    out = {
        "metric": "pseudo-label frames/sec on nuScenes-shaped sweeps", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r["dt"] / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {args.frames} frames/GPU x ({cfg.n_points}x{cfg.n_sweeps} pts, {cfg.n_cams} cams, "
                               f"{cfg.n_masks} masks {cfg.width}x{cfg.height}), masks resident as {main_mode}",
                   "frames_per_gpu": args.frames, "points_per_frame": cfg.n_points * cfg.n_sweeps, "masks_per_frame": cfg.n_masks,
                   "mask_size": [cfg.width, cfg.height], "lane_points": args.lane_points, "mask_input": main_mode,
                   "batches_in_flight": depth,
                   "parallelism": f"frame-sharded x{world}, {depth} independent batches in flight per GPU, one RCCL gather of box records"},
        "roofline": roofline,
        "kernels": kernels,
        "frame_alg_bytes": int(alg["frame_total"] // hb.n_frames),
        # the whole pass against the HBM roof: SURVEY 8(d)'s end-to-end algorithmic bytes per frame x frames/s
        "pass_roofline": {"bound": "hbm", "achieved": round(alg["frame_total"] / hb.n_frames * value / world / 1e9, 1), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(alg["frame_total"] / hb.n_frames * value / world / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "per GPU; ALG_FRAME with the masks read in the form they are resident in"},
        "boxes_per_step": r["n_boxes"], "in_mask_points_per_step": r["sum_hits"], "max_points_in_a_mask": r["max_hits"],
        "gen_seconds": round(t_gen, 1),
    }
    for mode in modes[1:]:
        o = results[mode]
        out[f"value_{mode}_masks"] = round(frames_total / o["dt"], 1)
        out[f"stage_ms_{mode}_masks"] = {k: round(v, 4) for k, v in dict(o["stage_ms"], project=o["project_alone_ms"]).items()}
    if world == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(frames, lanes, frame_lane, hb, args.cpu_sample)
        workers = min(args.cpu_workers, os.cpu_count() or 1)
        if workers > 1:
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.config, args.set, args.lane_points, workers,
                                                                       max(8, args.cpu_sample // 4))
            except Exception as exc:        # a reported extra: never let it break the bench line
                out["cpu_baseline_all_cores"] = {"error": repr(exc)}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
