#!/usr/bin/env python3
"""Drop-in for the reference's src/nuscenes/linear_matching.py: fuses the lifted pseudo-labels with a SAM3D result
file.  Run from this directory with no arguments: reads ../../outputs/nuscenes/ablation_1new_val_0_150_detic.json
and ../../outputs/sam3d_results_nusc_val.json (the reference's module constants :125-129), writes
../../outputs/matched_pseudolabels_nusc_train_0322.json per alpha and the best alpha's file to
../../outputs/best_matched_pseudolabels_nusc_train_0322.json (:452,:486).  The per-sample box matching runs on the
MI355X (cm3d_bev_match); each alpha is scored with this package's evaluation (cm3d_amd.eval_detection, all classes,
no drivable filtering, :455-478).  Paths and the table version are overridable through CM3D_* variables."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

from cm3d_amd import eval_detection as ev, fusion, nusc_io  # noqa: E402

INPUT_PATH = os.environ.get("CM3D_INPUT_PATH", "../../data/nuScenes")                                             # :125
VER_NAME = os.environ.get("CM3D_VER_NAME", "v1.0-trainval")                                                       # :145
pred_load_dir = os.environ.get("CM3D_PRED_JSON", "../../outputs/nuscenes/ablation_1new_val_0_150_detic.json")     # :128
sam3d_load_dir = os.environ.get("CM3D_SAM3D_JSON", "../../outputs/sam3d_results_nusc_val.json")                   # :129
OUT_PATH = os.environ.get("CM3D_MATCHED_JSON", "../../outputs/matched_pseudolabels_nusc_train_0322.json")         # :452
BEST_PATH = os.environ.get("CM3D_BEST_JSON", "../../outputs/best_matched_pseudolabels_nusc_train_0322.json")      # :486
EVAL_OUTPUT_DIR = os.environ.get("CM3D_OUTPUT_DIR", "../../outputs/nuscenes/")                                    # :459


def main():
    with open(sam3d_load_dir) as f:
        sam3d_objects = json.load(f)
    with open(pred_load_dir) as f:
        pred_objects = json.load(f)
    tables = nusc_io.NuscTables(VER_NAME, INPUT_PATH)
    cfg = ev.config_factory('detection_cvpr_2019')
    scenes = os.environ.get("CM3D_EVAL_SCENES")
    scenes = set(scenes.split(",")) if scenes else None

    def evaluate(path):
        de = ev.DetectionEval(tables, cfg, path, scenes, EVAL_OUTPUT_DIR, drivable_filtering=False, object_only=False, verbose=False)
        return de.main()["mean_ap"]

    alpha, score = fusion.grid_search(pred_objects, sam3d_objects, evaluate, OUT_PATH, BEST_PATH)
    print(f"best alpha {alpha}, mAP {score}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
