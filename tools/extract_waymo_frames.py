#!/usr/bin/env python3
"""Writes the per-frame `.npz` files that src/waymo/2d_to_3d.py reads, from Waymo Open Dataset TFRecords.
Needs waymo_open_dataset + TensorFlow (third-party; not part of this repository's image).  It performs exactly
the extraction steps of the reference's src/waymo/2d_to_3d.py:444-479,513-518,459-468 and nothing else."""
import os
import sys

import numpy as np


def main(tfrecord, out_dir):
    import tensorflow.compat.v1 as tf
    from waymo_open_dataset import dataset_pb2
    from waymo_open_dataset.utils import frame_utils
    tf.enable_eager_execution()
    scene = os.path.splitext(os.path.basename(tfrecord))[0]
    os.makedirs(os.path.join(out_dir, scene), exist_ok=True)
    for fnum, data in enumerate(tf.data.TFRecordDataset(tfrecord, compression_type="")):
        frame = dataset_pb2.Frame()
        frame.ParseFromString(bytearray(data.numpy()))
        ri, cp, _, top_pose = frame_utils.parse_range_image_and_camera_projection(frame)
        pts, _ = frame_utils.convert_range_image_to_point_cloud(frame, ri, cp, top_pose, 0, False)
        cals = sorted(frame.context.camera_calibrations, key=lambda c: c.name)
        rec = dict(points=np.asarray(pts[0], np.float32), extrinsics=np.array([list(c.extrinsic.transform) for c in cals]),
                   intrinsics=np.array([list(c.intrinsic) for c in cals]), pose=np.array(frame.pose.transform),
                   timestamp_micros=np.int64(frame.timestamp_micros), context_name=np.str_(frame.context.name))
        if fnum == 0:
            polys = [np.array([[p.x, p.y, p.z] for p in f.lane.polyline]) for f in frame.map_features if f.HasField("lane")]
            rec["lanes"] = np.vstack(polys)
            rec["lane_off"] = np.concatenate([[0], np.cumsum([len(p) for p in polys])])
        np.savez_compressed(os.path.join(out_dir, scene, f"{fnum}_frame.npz"), **rec)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
