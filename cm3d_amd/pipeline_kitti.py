"""KITTI entry point of the lifting path (reference src/kitti/2d_to_3d.py:896-1571, stage 1 as intended:
the committed script exits at a debug print, :1528).  Reads `<INPUT_DIR>/<f>_masks.pkl|_data.json`,
`<KITTI>/training/{velodyne/%06d.bin, calib/%06d.txt}` and writes `training/pred/%06d.txt` (with score)
and `training/pseudo/%06d.txt` (:1025-1036, :879-885)."""
import argparse
import glob
import json
import os
import pickle
import time

import torch

from . import kitti as kt
from . import lifting

INPUT_DIR = "../../mask_outputs/kitti-detic/"
KITTI_DIR = "../../data/kitti/"


def main(argv=None):
    ap = argparse.ArgumentParser(description="CM3D 2D->3D lifting (KITTI), MI355X path")
    ap.add_argument("--kitti-dir", default=os.environ.get("CM3D_KITTI_DIR", KITTI_DIR))
    ap.add_argument("--mask-dir", default=os.environ.get("CM3D_INPUT_DIR", INPUT_DIR))
    ap.add_argument("--priors", default="cfg/shape_priors_chatgpt.json")
    ap.add_argument("--ratio", type=float, default=kt.RATIO)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args(argv)
    t0 = time.time()
    pri = json.load(open(args.priors)) if os.path.exists(args.priors) else dict(lifting.SHAPE_PRIORS_CHATGPT)
    classes = lifting.ClassTable.nuscenes(pri)
    tr = os.path.join(args.kitti_dir, "training")
    pred_dir, pseudo_dir = os.path.join(tr, "pred"), os.path.join(tr, "pseudo")
    os.makedirs(pred_dir, exist_ok=True)
    os.makedirs(pseudo_dir, exist_ok=True)
    nums = sorted(int(os.path.basename(p).split("_")[0]) for p in glob.glob(os.path.join(args.mask_dir, "*_masks.pkl")))
    eng = lifting.LiftEngine("cuda:0", classes=classes)
    lane = [[0.0, 0.0, 0.0]]                   # stage 1 of KITTI uses no lanes; the engine still wants a table
    n_lines = 0
    for b0 in range(0, len(nums), args.batch):
        frames = []
        for fnum in nums[b0:b0 + args.batch]:
            with open(os.path.join(args.mask_dir, f"{fnum}_masks.pkl"), "rb") as f:
                rles = pickle.load(f)
            with open(os.path.join(args.mask_dir, f"{fnum}_data.json")) as f:
                data = json.load(f)
            for kind in (pred_dir, pseudo_dir):          # :1025-1036: files are recreated empty
                open(os.path.join(kind, f"{fnum:06d}.txt"), "w").close()
            if not rles:
                continue
            frames.append(kt.frame_from_files(fnum, os.path.join(tr, "velodyne", f"{fnum:06d}.bin"),
                                              os.path.join(tr, "calib", f"{fnum:06d}.txt"), rles, data["labels"],
                                              data["detection_scores"], args.ratio))
        by_size = {}
        for f in frames:
            by_size.setdefault((f.width, f.height), []).append(f)
        for _, fs in sorted(by_size.items()):
            hb = lifting.pack_frames(fs, [lane], [0] * len(fs), classes)
            eng.upload(hb)
            eng.run(masks="rle")
            torch.cuda.synchronize()
            res = eng.download()
            for i, f in enumerate(fs):
                pred, pseudo = kt.labels_of_frame(hb, res, i, classes, pri)
                with open(os.path.join(pred_dir, f"{f.token}.txt"), "a") as fh:
                    fh.writelines(pred)
                with open(os.path.join(pseudo_dir, f"{f.token}.txt"), "a") as fh:
                    fh.writelines(pseudo)
                n_lines += len(pred)
    print(f"wrote {n_lines} labels for {len(nums)} frames in {time.time() - t0:.2f} s")
    return 0
