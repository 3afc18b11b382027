#!/usr/bin/env python3
"""Drop-in for the reference's src/nuscenes/2d_to_3d.py: run from this directory with no arguments,
reads ../../mask_outputs/nuscenes-detic/<scene>/<f>_{masks.pkl,data.json} and ../../data/nuScenes/,
writes ../../outputs/nuscenes/pseudolabels_minival.json.  The per-frame work runs on the MI355X."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

from cm3d_amd.pipeline_nuscenes import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
