"""Waymo entry point of the lifting path: what `python 2d_to_3d.py` does in the reference's
src/waymo/ (2d_to_3d.py:394-1306), with the per-frame work on the GPU.

Inputs: the reference's mask files `<INPUT_DIR>/<scene>/<f>_masks.pkl|_data.json` plus, per frame, the
quantities the reference pulls out of the TFRecord with waymo_open_dataset / TensorFlow (third-party,
not in this image): `<FRAMES_DIR>/<scene>/<f>_frame.npz` with
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

INPUT_DIR = "../../mask_outputs/waymo-detic/"            # reference module constants (src/waymo/2d_to_3d.py)
FRAMES_DIR = "../../data/waymo_extracted/"
OUTPUT_PATH = "../../outputs/waymo/pred_detic.bin"


def load_scene(frames_dir, mask_dir, scene):
    """-> (frames, lane_table) of one scene; frames without mask files are skipped like :450-455."""
    files = sorted(glob.glob(os.path.join(frames_dir, scene, "*_frame.npz")), key=lambda p: int(os.path.basename(p).split("_")[0]))
    frames, lane_table = [], None
    for path in files:
        fnum = int(os.path.basename(path).split("_")[0])
        z = np.load(path, allow_pickle=False)
        if fnum == 0 or lane_table is None and "lanes" in z:
            if "lanes" in z:
                off = z["lane_off"]
                lane_table = np.vstack([wm.get_yaws_from_lane_coords(z["lanes"][off[i]:off[i + 1]]) for i in range(len(off) - 1)])
        mp = os.path.join(mask_dir, scene, f"{fnum}_masks.pkl")
        dp = os.path.join(mask_dir, scene, f"{fnum}_data.json")
        if not (os.path.exists(mp) and os.path.exists(dp)):
            continue
        with open(mp, "rb") as f:
            rles = pickle.load(f)
        with open(dp) as f:
            data = json.load(f)
        if not rles:
            continue
        W, H = rles[0]["size"]
        cams = [(z["extrinsics"][c], z["intrinsics"][c]) for c in range(z["extrinsics"].shape[0])]
        frames.append(wm.frame_from_extracted(f"{scene}:{fnum}", z["points"], cams, rles, data["labels"], data["detection_scores"],
                                              data["cam_nums"], z["pose"], W, H, int(z["timestamp_micros"]), str(z["context_name"])))
    return frames, lane_table


def lift_scene(eng, frames, lane_table, classes, masks="rle"):
    objs = []
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
        res = eng.download(full=False)      # objects need boxes and flags only
        objs.append((hb, res, [(f.context_name, f.timestamp_micros) for f in fs]))
    encoded = []
    for hb, res, meta in objs:
        encoded.extend(wm.objects_from_results(hb, res, classes, meta))
    return encoded


def main(argv=None):
    ap = argparse.ArgumentParser(description="CM3D 2D->3D lifting (Waymo), MI355X path")
    ap.add_argument("--frames-dir", default=os.environ.get("CM3D_WAYMO_FRAMES", FRAMES_DIR))
    ap.add_argument("--mask-dir", default=os.environ.get("CM3D_INPUT_DIR", INPUT_DIR))
    ap.add_argument("--output", default=os.environ.get("CM3D_OUTPUT_PATH", OUTPUT_PATH))
    ap.add_argument("--scenes", default=os.environ.get("CM3D_SCENES", ""))
    ap.add_argument("--priors", default="cfg/shape_priors_chatgpt.json")
    ap.add_argument("--masks", default="rle", choices=["rle", "dense"])
    args = ap.parse_args(argv)
    t0 = time.time()
    rank, world, local_rank = cdist.init_from_env()
    scenes = [s for s in args.scenes.split(",") if s] or sorted(os.listdir(args.frames_dir))
    pri = json.load(open(args.priors)) if os.path.exists(args.priors) else None
    classes = lifting.ClassTable.waymo(pri)
    eng = lifting.LiftEngine(f"cuda:{local_rank}", classes=classes)
    lo, hi = cdist.shard_range(len(scenes), rank, world)
    mine = []
    for scene in scenes[lo:hi]:
        frames, lanes = load_scene(args.frames_dir, args.mask_dir, scene)
        if frames:
            if lanes is None:
                raise FileNotFoundError(f"{scene}: no lane polylines (frame 0)")
            mine.extend(lift_scene(eng, frames, lanes, classes, args.masks))
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        torch.distributed.gather_object(mine, gathered, dst=0)
        if rank != 0:
            return 0
        mine = [o for part in gathered for o in part]
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    with open(args.output, "wb") as f:
        f.write(wm.encode_objects(mine))
    print(f"wrote {len(mine)} objects in {time.time() - t0:.2f} s")
    return 0
