"""Shared helpers of the parity tests: run frames through the CPU oracle in the
reference's execution order and return arrays shaped like LiftEngine.download()."""
import numpy as np


def oracle_batch(orc, frames, lane_tables, frame_lane, hb):
    pts_all, pt_off = [], [0]
    hit_idx, hit_off = [], [0]
    medoid_pos, centroid = [], []
    lane_idx, lane_dist, box, flags, bbox = [], [], [], [], []
    for fi, fr in enumerate(frames):
        halfw = orc.EGO_HALFW_F32 if hb.ego_box else np.float32(0.0)
        pts = np.concatenate([orc.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24], halfw)
                              for r, x in zip(fr.sweeps_raw, fr.sweep_xf)], 0)
        pts_all.append(pts)
        pt_off.append(pt_off[-1] + pts.shape[0])
        masks = [orc.rle_decode(r).T for r in fr.rles]     # (H,W) image layout, like depth_images[i]
        idx_lists, med, cent = orc.lift_frame_reference_order(pts, fr.cams, masks, fr.cam_nums)
        for m in masks:
            er = orc.erode3x3(m)
            ys, xs = np.nonzero(er)
            bbox.append([xs.min(), ys.min(), xs.max(), ys.max()] if xs.size else [0x7FFFFFFF, 0x7FFFFFFF, -1, -1])
        for il in idx_lists:
            hit_idx.append(il)
            hit_off.append(hit_off[-1] + il.size)
        medoid_pos.append(med)
        centroid.append(np.where(np.isnan(cent), 0, cent))
        m0, m1 = hb.mask_off[fi], hb.mask_off[fi + 1]
        if hb.pose_rt is not None:      # Waymo (a17)
            s2 = orc.stage2_frame_waymo(cent, med, hb.class_id[m0:m1], hb.score[m0:m1], lane_tables[frame_lane[fi]],
                                        hb.pose_rt[fi], hb.pose_inv[fi])
            s2["rotation"] = np.zeros((m1 - m0, 4))
            s2["rotation"][:, 0] = s2["heading"]
            s2["rotation"][~s2["valid"], 0] = 1.0
        else:
            s2 = orc.stage2_frame(cent, med, hb.class_id[m0:m1], hb.score[m0:m1], lane_tables[frame_lane[fi]], fr.ego_xyz)
        lane_idx.append(s2["lane_idx"]); lane_dist.append(s2["lane_dist"])
        b = np.zeros((m1 - m0, 10))
        b[:, 0:3] = np.where(s2["valid"][:, None], s2["translation"], 0.0)
        b[:, 3] = np.where(s2["valid"], s2["rotation"][:, 0], 1.0)
        b[:, 4] = np.where(s2["valid"], s2["rotation"][:, 3], 0.0)
        b[:, 5] = s2["yaw"]
        b[:, 6] = np.where(s2["valid"], s2["lane_dist"], 0.0)
        b[:, 7] = hb.score[m0:m1]
        b[:, 8] = hb.class_id[m0:m1]
        b[:, 9] = s2["valid"].astype(np.int32) | (s2["keep"].astype(np.int32) << 1)
        box.append(b)
        flags.append(s2["valid"].astype(np.int32) | (s2["keep"].astype(np.int32) << 1))
    return dict(points=np.concatenate(pts_all, 0), pt_off=np.array(pt_off, np.int32),
                hit_idx=np.concatenate(hit_idx) if hit_idx else np.zeros(0, np.int32), hit_off=np.array(hit_off, np.int32),
                medoid_pos=np.concatenate(medoid_pos), centroid=np.concatenate(centroid, 0).astype(np.float32),
                lane_idx=np.concatenate(lane_idx), lane_dist=np.concatenate(lane_dist), box=np.concatenate(box, 0),
                flags=np.concatenate(flags), bbox=np.array(bbox, np.int32))


def oracle_results(orc, frames, lane_tables, frame_lane, classes=None):
    """End-to-end oracle: frames -> {token: [box dict]} like lifting.box_records, through the
    reference-order oracle (stage 1, stage 2, NMS)."""
    from cm3d_amd import lifting
    classes = classes or lifting.ClassTable.nuscenes()
    live = [i for i, f in enumerate(frames) if len(f.rles) > 0]
    out = {f.token: [] for f in frames}
    if not live:
        return out
    sub = [frames[i] for i in live]
    fl = [frame_lane[i] for i in live]
    hb = lifting.pack_frames(sub, lane_tables, fl, classes)
    exp = oracle_batch(orc, sub, lane_tables, fl, hb)
    out.update(lifting.box_records(hb, exp, classes))
    return out


def g7r_scene(tmpdir):
    """The scene of golden G7r (tests/golden/gen_golden_chain.py): the tiny scene of G7 -- one scene, two frames, written and read
    back through the on-disk formats -- with the first three detections of every frame listed a second time at a lower score (the
    same mask, so the same centroid: the per-sample NMS of the reference must drop the copies).  Returns (frames, lane points)."""
    from cm3d_amd import nusc_io, synthetic as syn
    tiny = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmpdir), tiny, n_scenes=1, frames_per_scene=2)
    tables = nusc_io.NuscTables("v1.0-synth", dataroot)
    scene = tables.scene_by_name(names[0])
    frames = nusc_io.frames_of_scene(tables, scene, mask_dir, n_sweeps=3, ratio=tiny.ratio)
    for f in frames:
        f.rles = list(f.rles) + list(f.rles[:3])
        f.labels = list(f.labels) + list(f.labels[:3])
        f.cam_nums = list(f.cam_nums) + list(f.cam_nums[:3])
        f.scores = list(f.scores) + [0.5 * float(s) for s in f.scores[:3]]
    return frames, nusc_io.load_lane_points(dataroot, tables.location(scene))
