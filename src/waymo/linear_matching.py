#!/usr/bin/env python3
"""Drop-in for the reference's src/waymo/linear_matching.py: fuses the lifted Waymo pseudo-labels with a SAM3D Objects
file.  Run from this directory: reads ../../outputs/waymo/pseudolabels_waymo_0307_train_0_798.bin and
../../../SAM3D/pred_outputs/sam3d_outputs/waymo-train.bin (the reference's module constants :139-163), writes
../../outputs/waymo/matched_pseudolabels_waymo_train_0310.bin per alpha and the best alpha's file to
best_matched_pseudolabels_waymo_train_0310.bin (:470,:541).  The per-frame box matching runs on the MI355X
(cm3d_bev_match).  Each alpha is scored like the reference does, by waymo-open-dataset's compute_detection_metrics_main
(:476-537), an external binary: give its path in CM3D_WAYMO_METRICS_BIN and the ground-truth file in CM3D_WAYMO_GT_BIN."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

from cm3d_amd import fusion, waymo as wm  # noqa: E402

OUTPUT_DIR = os.environ.get("CM3D_OUTPUT_DIR", "../../outputs/waymo/")                                                   # :142
PRED_BIN = os.environ.get("CM3D_PRED_BIN", "../../outputs/waymo/pseudolabels_waymo_0307_train_0_798.bin")                # :161
SAM3D_BIN = os.environ.get("CM3D_SAM3D_BIN", "../../../SAM3D/pred_outputs/sam3d_outputs/waymo-train.bin")                # :151
METRICS_BIN = os.environ.get("CM3D_WAYMO_METRICS_BIN",
                             os.path.expanduser("~/mmdetection3d/mmdet3d/evaluation/functional/waymo_utils/compute_detection_metrics_main"))
GT_BIN = os.environ.get("CM3D_WAYMO_GT_BIN", "../../data/waymo-v1.4.2/waymo_format/gt-training.bin")                     # :478


def main():
    with open(SAM3D_BIN, "rb") as f:
        sam3d = wm.decode_objects(f.read())
    with open(PRED_BIN, "rb") as f:
        pred = wm.decode_objects(f.read())

    def evaluate(path):
        text = subprocess.check_output([METRICS_BIN, path, GT_BIN]).decode("utf-8")
        print(text)
        return fusion.parse_waymo_metrics(text)[1]

    alpha, score = fusion.waymo_grid_search(pred, sam3d, evaluate, os.path.join(OUTPUT_DIR, "matched_pseudolabels_waymo_train_0310.bin"),
                                            os.path.join(OUTPUT_DIR, "best_matched_pseudolabels_waymo_train_0310.bin"))
    print(f"best alpha {alpha}, Overall/L2 mAP {score}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
