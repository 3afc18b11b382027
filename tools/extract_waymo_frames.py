#!/usr/bin/env python3
"""Writes the per-frame `.npz` files that src/waymo/2d_to_3d.py reads when it is not given the TFRecords themselves, from Waymo
Open Dataset TFRecords.  Needs waymo_open_dataset + TensorFlow (third-party; not part of this repository's image).  The
extraction itself is cm3d_amd.waymo.frame_records_from_tfrecord -- the very function the entry point calls when it reads the
TFRecords directly (--tfrecords): the steps of the reference's src/waymo/2d_to_3d.py:444-479,513-518,459-468 and nothing else."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(tfrecord, out_dir, scene=None):
    from cm3d_amd import waymo as wm
    scene = scene or os.path.splitext(os.path.basename(tfrecord))[0]
    os.makedirs(os.path.join(out_dir, scene), exist_ok=True)
    for fnum, rec in wm.frame_records_from_tfrecord(tfrecord):
        np.savez_compressed(os.path.join(out_dir, scene, f"{fnum}_frame.npz"), **rec)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
