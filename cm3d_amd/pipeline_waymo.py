"""Waymo entry point of the lifting path: what `python 2d_to_3d.py` does in the reference's
src/waymo/ (2d_to_3d.py:394-1306), with the per-frame work on the GPU.

Inputs: the reference's mask files `<INPUT_DIR>/<scene>/<f>_masks.pkl|_data.json` plus, per frame, the
quantities the reference pulls out of the TFRecord with waymo_open_dataset / TensorFlow (third-party).  Two routes to them,
giving the same kernel inputs (tests/test_host_logic.py holds them against each other with a stand-in devkit):
  * `--tfrecords DIR` (default: the reference's INPUT_PATH) where the devkit is installed: the entry point reads the
    TFRecords itself, scene = file name, exactly like the reference (src/waymo/2d_to_3d.py:424-479) --
    cm3d_amd.waymo.frame_records_from_tfrecord;
  * otherwise `<FRAMES_DIR>/<scene>/<f>_frame.npz` (tools/extract_waymo_frames.py writes them with that same function) with
    points (N,3) float32   TOP-lidar first returns in the vehicle frame   (:472-479)
    extrinsics (5,16), intrinsics (5,9)   camera calibrations in camera-name order 1..5 (:513-518)
    pose (16,), timestamp_micros, context_name
    lanes (L,3) float64  x,y,z of all lane polylines concatenated + lane_off (n_lanes+1)  [frame 0 only, :459-468]
`tools/extract_waymo_frames.py` writes these files where the devkit is installed.
Output: a serialised metrics_pb2.Objects (:1300-1305).
"""
import argparse
import glob
import json
import os
import pickle
import time

import numpy as np
import torch

from . import dist as cdist
from . import lifting
from . import waymo as wm

INPUT_PATH = "../../data/waymo-v1.4.2/waymo_format/training/"      # reference module constants (src/waymo/2d_to_3d.py:47,64-65)
INPUT_DIR = "../../mask_outputs/waymo-detic/"
FRAMES_DIR = "../../data/waymo_extracted/"
OUTPUT_PATH = "../../outputs/waymo/pred_detic.bin"


def _npz_records(frames_dir, scene):
    files = sorted(glob.glob(os.path.join(frames_dir, scene, "*_frame.npz")), key=lambda p: int(os.path.basename(p).split("_")[0]))
    for path in files:
        yield int(os.path.basename(path).split("_")[0]), np.load(path, allow_pickle=False)


def load_scene(frames_dir, mask_dir, scene, tfrecord=None):
    """-> (frames, lane_table) of one scene; frames without mask files are skipped like :450-455.
    tfrecord: path of the scene's TFRecord -> read through the devkit (cm3d_amd.waymo.frame_records_from_tfrecord);
    None -> the extracted `<frames_dir>/<scene>/<f>_frame.npz` files."""
    def has_masks(fnum):
        return os.path.exists(os.path.join(mask_dir, scene, f"{fnum}_masks.pkl")) and os.path.exists(os.path.join(mask_dir, scene, f"{fnum}_data.json"))
    records = wm.frame_records_from_tfrecord(tfrecord, has_masks) if tfrecord else _npz_records(frames_dir, scene)
    frames, lane_table = [], None
    for fnum, z in records:
        if lane_table is None and "lanes" in z:
            off = z["lane_off"]
            if len(off) > 1:
                lane_table = np.vstack([wm.get_yaws_from_lane_coords(z["lanes"][off[i]:off[i + 1]]) for i in range(len(off) - 1)])
        if not has_masks(fnum):
            continue
        with open(os.path.join(mask_dir, scene, f"{fnum}_masks.pkl"), "rb") as f:
            rles = pickle.load(f)
        with open(os.path.join(mask_dir, scene, f"{fnum}_data.json")) as f:
            data = json.load(f)
        if not rles:
            continue
        W, H = rles[0]["size"]
        cams = [(z["extrinsics"][c], z["intrinsics"][c]) for c in range(z["extrinsics"].shape[0])]
        frames.append(wm.frame_from_extracted(f"{scene}:{fnum}", z["points"], cams, rles, data["labels"], data["detection_scores"],
                                              data["cam_nums"], z["pose"], W, H, int(z["timestamp_micros"]), str(z["context_name"])))
    return frames, lane_table


def lift_scene(eng, frames, lane_table, classes, masks="rle", scene_index=0):
    """One scene through the hot path; returns its kept-box records as device tensors (lifting.kept_box_records with
    column 5 = scene_index, column 6 = the frame's number inside the scene), one tensor per image size."""
    recs = []
    by_size = {}
    for f in frames:
        by_size.setdefault((f.width, f.height), []).append(f)     # Waymo has two image sizes (:520-523)
    for _, fs in sorted(by_size.items()):
        hb = lifting.pack_frames(fs, [lane_table], [0] * len(fs), classes)
        eng.upload(hb)
        if masks == "dense":
            eng.decode_masks_dense()
        eng.run(masks=masks)
        torch.cuda.synchronize()
        eng.check_status()
        ids = np.array([[scene_index, int(f.token.rsplit(":", 1)[1])] for f in fs], np.float64)
        recs.append(lifting.kept_box_records(eng.b, ids))
    return recs


def objects_from_records(rec, scenes, frames_dir, classes, meta=None):
    """Gathered records -> encoded metrics_pb2.Object payloads in the reference's order (scenes in order, frames in
    order, kept boxes in mask order; src/waymo/2d_to_3d.py:1034-1065,1262-1297).  The frame's context name and timestamp
    (:1050-1051) come from its extracted-frame file; the records only carry (scene index, frame number)."""
    rec = np.asarray(rec, np.float64).reshape(-1, 10)
    order = np.lexsort((np.arange(len(rec)), rec[:, lifting.REC_FRAME_B], rec[:, lifting.REC_FRAME_A]))     # stable inside a frame
    meta, out = dict(meta or {}), []           # (scene index, frame number) -> (context name, timestamp): what this rank saw ...
    for r in rec[order]:
        key = (int(r[lifting.REC_FRAME_A]), int(r[lifting.REC_FRAME_B]))
        if key not in meta:                    # ... and, for the frames of other ranks, the frame's extracted file
            z = np.load(os.path.join(frames_dir, scenes[key[0]], f"{key[1]}_frame.npz"), allow_pickle=False)
            meta[key] = (str(z["context_name"]), int(z["timestamp_micros"]))
        ctx, ts = meta[key]
        ci = int(r[8])
        pr = classes.prior_wlh[ci]
        out.append(wm.encode_object(r[0:3], length=pr[1], width=pr[0], height=pr[2], heading=r[3],
                                    type_id=wm.WAYMO_TYPE[classes.out_names[ci]], score=r[7], context_name=ctx, timestamp_micros=ts))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="CM3D 2D->3D lifting (Waymo), MI355X path")
    ap.add_argument("--frames-dir", default=os.environ.get("CM3D_WAYMO_FRAMES", FRAMES_DIR))
    ap.add_argument("--tfrecords", default=os.environ.get("CM3D_WAYMO_TFRECORDS", INPUT_PATH),
                    help="directory of Waymo TFRecords (read through waymo_open_dataset when it is installed and the directory exists; "
                         "otherwise the extracted <f>_frame.npz files of --frames-dir are used)")
    ap.add_argument("--mask-dir", default=os.environ.get("CM3D_INPUT_DIR", INPUT_DIR))
    ap.add_argument("--output", default=os.environ.get("CM3D_OUTPUT_PATH", OUTPUT_PATH))
    ap.add_argument("--scenes", default=os.environ.get("CM3D_SCENES", ""))
    ap.add_argument("--priors", default="cfg/shape_priors_chatgpt.json")
    ap.add_argument("--masks", default="rle", choices=["rle", "dense"])
    args = ap.parse_args(argv)
    t0 = time.time()
    rank, world, local_rank = cdist.init_from_env()
    if os.environ.get("CM3D_SINGLE_DEVICE"):      # rehearsal of the N>1 path on a one-GPU box (with CM3D_DIST_BACKEND=gloo)
        local_rank = 0
    use_tf = os.path.isdir(args.tfrecords) and wm.devkit_available()
    # scene = TFRecord file name, like the reference (:424-436); with extracted frames: the directory name
    scenes = [s for s in args.scenes.split(",") if s] or sorted(os.listdir(args.tfrecords if use_tf else args.frames_dir))
    pri = json.load(open(args.priors)) if os.path.exists(args.priors) else None
    classes = lifting.ClassTable.waymo(pri)
    eng = lifting.LiftEngine(f"cuda:{local_rank}", classes=classes)
    lo, hi = cdist.shard_range(len(scenes), rank, world)
    mine, meta = [], {}
    for si in range(lo, hi):
        frames, lanes = load_scene(args.frames_dir, args.mask_dir, scenes[si], os.path.join(args.tfrecords, scenes[si]) if use_tf else None)
        for f in frames:
            meta[(si, int(f.token.rsplit(":", 1)[1]))] = (f.context_name, f.timestamp_micros)
        if frames:
            if lanes is None:
                raise FileNotFoundError(f"{scenes[si]}: no lane polylines (frame 0)")
            mine.extend(lift_scene(eng, frames, lanes, classes, args.masks, scene_index=si))
    rec = torch.cat(mine, 0) if mine else torch.zeros(0, 10, dtype=torch.float64, device=eng.dev)
    # the single exchange of the job: fixed-size box records -> rank 0 (also the path of a one-rank run)
    gathered = cdist.gather_records(rec, dst=0)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return 0
    if use_tf and world > 1:
        # the records carry (scene, frame) only; context name and timestamp of the other ranks' frames come from their TFRecords
        for si in [i for i in range(len(scenes)) if not (lo <= i < hi)]:
            for fnum, z in wm.frame_records_from_tfrecord(os.path.join(args.tfrecords, scenes[si]), lambda k: False):
                meta[(si, fnum)] = (str(z["context_name"]), int(z["timestamp_micros"]))
    mine = objects_from_records(torch.cat([g.cpu() for g in gathered], 0).numpy(), scenes, args.frames_dir, classes, meta)
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    with open(args.output, "wb") as f:
        f.write(wm.encode_objects(mine))
    print(f"wrote {len(mine)} objects in {time.time() - t0:.2f} s")
    return 0
