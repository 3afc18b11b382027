"""SAM3D fusion step (reference src/nuscenes/linear_matching.py, src/waymo/linear_matching.py).

The reference merges the lifted pseudo-labels ("pred", the output of 2d_to_3d.py) with the boxes of a SAM3D
run: per sample the two box sets are matched one to one on bird's-eye-view IoU >= 0.2 (Hungarian matcher of
waymo_open_dataset, :231-259), then for every alpha of a grid the SAM3D scores are rescaled by alpha, each
matched pair keeps the box of the higher score, unmatched boxes of both sides are kept, the merged file is
evaluated and the best alpha's file is kept (:263-491).

Here the matching of all samples is one GPU call (cm3d_bev_match, no CPU fallback); the dictionary work around
it is host Python as in the reference.  Line cites: src/nuscenes/linear_matching.py unless noted.
"""
import json
import os

import numpy as np

from . import ops

FUSED_META = {"use_camera": True, "use_lidar": True, "use_radar": False, "use_map": True, "use_external": False}   # :281-287


def heading_of(rotation):
    """:171,210 -- `R.from_quat(obj["rotation"]).as_euler('xyz')[0]`.  The stored quaternion is (w,x,y,z) and
    scipy reads (x,y,z,w), so a yaw-only box of yaw psi comes out with heading pi - psi; kept as is, the
    fused file carries these headings (:314-324)."""
    from scipy.spatial.transform import Rotation as R
    return R.from_quat(rotation).as_euler('xyz', degrees=False)[0]


def headings_of(rotations):
    """heading_of for a list of quaternions in one call (same scipy routine, element-wise identical)."""
    from scipy.spatial.transform import Rotation as R
    return R.from_quat(np.asarray(rotations, dtype=float)).as_euler('xyz', degrees=False)[:, 0]


def parse_results(results, zero_min_quirk=False):
    """:157-190 (SAM3D, zero_min_quirk=True) / :196-227 (pred): results[sample] -> per-sample lists of
    [x, y, bottom_z, size0, size1, size2, heading] and [attribute_name, score, velocity, detection_name];
    also the score range (a zero SAM3D score does not lower the minimum, :186-190)."""
    box_dict, supp_dict = {}, {}
    max_conf, min_conf = -1e7, 1e7
    # all headings of the file in one scipy call (the per-box call of the reference costs ~25 us x 10^6 boxes)
    quats = [obj["rotation"] for sample in results for obj in results[sample]]
    headings = iter(headings_of(quats)) if quats else iter(())
    for sample in results:
        box_dict.setdefault(sample, [])
        supp_dict.setdefault(sample, [])
        for obj in results[sample]:
            box_dict[sample].append(np.array([
                obj["translation"][0], obj["translation"][1], obj["translation"][2] - obj["size"][2] / 2,
                obj["size"][0], obj["size"][1], obj["size"][2], next(headings)], dtype=float))
            supp_dict[sample].append([obj["attribute_name"], obj["detection_score"], obj["velocity"], obj["detection_name"]])
            s = obj["detection_score"]
            if s > max_conf:
                max_conf = s
            if s < min_conf:
                if not zero_min_quirk or s != 0:
                    min_conf = s
    return box_dict, supp_dict, max_conf, min_conf


def match_samples(pred_box_dict, sam3d_box_dict, iou=0.2):
    """:231-259 -- per sample of the predictions that SAM3D also has: matched prediction / SAM3D indices.
    One cm3d_bev_match call for all samples."""
    pred_matches, sam3d_matches = {}, {}
    todo = []
    for ts in pred_box_dict:
        pred_matches.setdefault(ts, [])
        sam3d_matches.setdefault(ts, [])
        if ts not in sam3d_box_dict:                                     # :241-244 KeyError -> continue
            continue
        if len(pred_box_dict[ts]) == 0 or len(sam3d_box_dict[ts]) == 0:
            continue
        todo.append(ts)
    res = ops.bev_match([np.array(pred_box_dict[ts], dtype=float) for ts in todo],
                        [np.array(sam3d_box_dict[ts], dtype=float) for ts in todo], iou)
    for ts, (pi, gi, _) in zip(todo, res):
        pred_matches[ts] = [int(i) for i in pi]
        sam3d_matches[ts] = [int(i) for i in gi]
    return pred_matches, sam3d_matches


def alpha_grid(pred_min_conf, pred_max_conf, sam3d_min_conf, sam3d_max_conf, step=0.04):
    """:270-276."""
    return list(np.arange(pred_min_conf / sam3d_max_conf, pred_max_conf / sam3d_min_conf, step, dtype=float))


def yaw_quaternion(heading):
    """:315-324 `list(Quaternion(matrix=rot_matrix))` for rot_matrix = Rz(heading): pyquaternion 0.9.9's trace
    method restated for this matrix (w may be negative when cos(heading) < 0)."""
    c, s = float(np.cos(heading)), float(np.sin(heading))
    if c < -c:
        t = 1.0 - c - c + 1.0
        f = 0.5 / np.sqrt(t)
        return [float((s + s) * f), 0.0, 0.0, float(t * f)]
    t = 1.0 + c + c + 1.0
    f = 0.5 / np.sqrt(t)
    return [float(t * f), 0.0, 0.0, float((s + s) * f)]


def _box_dict(sample, box, name, score, attr):
    """:303-324."""
    return {
        "sample_token": sample,
        "translation": [float(box[0]), float(box[1]), float(box[2]) + float(box[5]) / 2],
        "size": [float(box[3]), float(box[4]), float(box[5])],
        "rotation": yaw_quaternion(box[6]),
        "velocity": [0, 0],
        "detection_name": name,
        "detection_score": score,
        "attribute_name": attr,
    }


def fuse(pred_box_dict, pred_supp_dict, sam3d_box_dict, sam3d_supp_dict, pred_matches, sam3d_matches, alpha):
    """:280-442 for one alpha.  Returns (matched_objects, counters).  A SAM3D sample without predictions has no
    match list in the reference (its loop would raise KeyError at :337); here it counts as unmatched."""
    out = {"meta": dict(FUSED_META), "results": {}}
    res = out["results"]
    n = dict(num_samples=0, num_pred_boxes=0, num_sam3d_boxes=0, num_sam3d_samples=0, num_matched_boxes=0)
    for ts in pred_box_dict:                                                         # :298-332
        for i, b in enumerate(pred_box_dict[ts]):
            if i in pred_matches[ts]:
                continue
            s = pred_supp_dict[ts][i]
            res.setdefault(ts, []).append(_box_dict(ts, b, s[3], s[1], s[0]))
            n["num_pred_boxes"] += 1
        n["num_samples"] += 1
    for ts in sam3d_box_dict:                                                        # :335-369
        matched = sam3d_matches.get(ts, [])
        for i, b in enumerate(sam3d_box_dict[ts]):
            if i in matched:
                continue
            s = sam3d_supp_dict[ts][i]
            res.setdefault(ts, []).append(_box_dict(ts, b, s[3], float(np.clip(s[1] * alpha, 0, 1)), s[0]))
            n["num_sam3d_boxes"] += 1
        n["num_sam3d_samples"] += 1
    for ts in pred_matches:                                                          # :374-442
        for k, pid in enumerate(pred_matches[ts]):
            sid = sam3d_matches[ts][k]
            sam3d_score = sam3d_supp_dict[ts][sid][1] * alpha
            pred_score = pred_supp_dict[ts][pid][1]
            ps = pred_supp_dict[ts][pid]
            if sam3d_score > pred_score:
                d = _box_dict(ts, sam3d_box_dict[ts][sid], ps[3], float(np.clip(sam3d_score, 0, 1)), ps[0])
            else:
                d = _box_dict(ts, pred_box_dict[ts][pid], ps[3], ps[1], ps[0])
            res.setdefault(ts, []).append(d)
            n["num_matched_boxes"] += 1
    return out, n


def grid_search(pred_objects, sam3d_objects, evaluate, out_path, best_path, iou=0.2, verbose=True):
    """:142-491: parse both files, match once, then for every alpha write the fused file to out_path, score
    it with evaluate(out_path) -> mean AP and keep the best one in best_path.  Returns (best_alpha, best_score)."""
    sam3d_box, sam3d_supp, sam3d_max, sam3d_min = parse_results(sam3d_objects["results"], zero_min_quirk=True)
    pred_box, pred_supp, pred_max, pred_min = parse_results(pred_objects["results"])
    pred_matches, sam3d_matches = match_samples(pred_box, sam3d_box, iou)
    best_alpha, best_score = 0, -1
    for alpha in alpha_grid(pred_min, pred_max, sam3d_min, sam3d_max):
        fused, counts = fuse(pred_box, pred_supp, sam3d_box, sam3d_supp, pred_matches, sam3d_matches, alpha)
        if verbose:
            for k, v in counts.items():
                print(k, v)
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(fused, f)
        map_score = float(evaluate(out_path))
        if map_score > best_score:
            best_score, best_alpha = map_score, alpha
            with open(best_path, "w") as f:
                json.dump(fused, f)
        if verbose:
            print(f"Curr Score: {map_score},  Curr Alpha: {alpha}")
            print(f"Best Score: {best_score}, Best Alpha: {best_alpha}")
            print("-" * 80)
    return best_alpha, best_score


# ------------------------------------------------------------------ Waymo (src/waymo/linear_matching.py)
def waymo_parse(objs, zero_min_quirk=False):
    """src/waymo/linear_matching.py:165-247 on decoded Objects (cm3d_amd.waymo.decode_objects): per (context, timestamp)
    the boxes [cx, cy, bottom_z, length, width, height, heading] and the objects themselves; score range as in
    parse_results."""
    boxes, supp = {}, {}
    max_conf, min_conf = -1e7, 1e7
    for o in objs:
        k = (o["context_name"], o["timestamp_micros"])
        c = o["center"]
        boxes.setdefault(k, []).append(np.array([c[0], c[1], c[2] - o["height"] / 2, o["length"], o["width"], o["height"], o["heading"]],
                                                dtype=float))
        supp.setdefault(k, []).append(o)
        s = o["score"]
        if s > max_conf:
            max_conf = s
        if s < min_conf and (not zero_min_quirk or s != 0):
            min_conf = s
    return boxes, supp, max_conf, min_conf


def waymo_alpha_grid(pred_min_conf, pred_max_conf, sam3d_min_conf, sam3d_max_conf, step=0.04):
    """src/waymo/linear_matching.py:322-328: the grid runs downwards and skips its three largest values."""
    a = np.arange(pred_min_conf / sam3d_max_conf, pred_max_conf / sam3d_min_conf + step, step, dtype=float)
    return list(a)[::-1][3:]


def fuse_waymo(pb, ps, sb, ss, pm, sm, alpha):
    """src/waymo/linear_matching.py:335-470 for one alpha: unmatched predictions, unmatched SAM3D boxes (score x alpha,
    clipped), then per matched pair the box of the higher score under the prediction's id and type.  Returns the
    encoded metrics_pb2.Object payloads (cm3d_amd.waymo.encode_objects serialises them)."""
    from . import waymo as wm

    def enc(k, b, src, score):
        return wm.encode_object([float(b[0]), float(b[1]), float(b[2]) + float(b[5]) / 2], length=float(b[3]), width=float(b[4]),
                                height=float(b[5]), heading=float(b[6]), type_id=src["type"], score=score, context_name=k[0],
                                timestamp_micros=k[1], object_id=src["id"])
    out = []
    for k in pb:
        out += [enc(k, b, ps[k][i], ps[k][i]["score"]) for i, b in enumerate(pb[k]) if i not in pm[k]]
    for k in sb:
        matched = sm.get(k, [])
        out += [enc(k, b, ss[k][i], float(np.clip(ss[k][i]["score"] * alpha, 0, 1))) for i, b in enumerate(sb[k]) if i not in matched]
    for k in pm:
        for j, pid in enumerate(pm[k]):
            sid = sm[k][j]
            s_score = ss[k][sid]["score"] * alpha
            if s_score > ps[k][pid]["score"]:
                out.append(enc(k, sb[k][sid], ps[k][pid], float(np.clip(s_score, 0, 1))))
            else:
                out.append(enc(k, pb[k][pid], ps[k][pid], ps[k][pid]["score"]))
    return out


def parse_waymo_metrics(text):
    """src/waymo/linear_matching.py:484-537: the output of compute_detection_metrics_main -> AP dict; 'Overall' entries
    are the mean over vehicle, pedestrian and cyclist.  Returns (ap_dict, ap_dict['Overall/L2 mAP'])."""
    keys = [f"{c}/{l} {m}" for c in ("Vehicle", "Pedestrian", "Sign", "Cyclist", "Overall") for l in ("L1", "L2") for m in ("mAP", "mAPH")]
    ap = {k: 0 for k in keys}
    map_splits, maph_splits = text.split('mAP '), text.split('mAPH ')
    for idx, key in enumerate(keys):
        split_idx = int(idx / 2) + 1
        if key.startswith("Overall"):
            continue
        ap[key] = float((map_splits if idx % 2 == 0 else maph_splits)[split_idx].split(']')[0])
    for l in ("L1", "L2"):
        for m in ("mAP", "mAPH"):
            ap[f"Overall/{l} {m}"] = (ap[f"Vehicle/{l} {m}"] + ap[f"Pedestrian/{l} {m}"] + ap[f"Cyclist/{l} {m}"]) / 3
    return ap, ap["Overall/L2 mAP"]


def waymo_grid_search(pred_objs, sam3d_objs, evaluate, out_path, best_path, iou=0.2, verbose=True):
    """src/waymo/linear_matching.py:165-547: match once (GPU), then per alpha write the fused Objects file to out_path, score
    it with evaluate(out_path) -> Overall/L2 mAP and keep the best in best_path.  Returns (best_alpha, best_score)."""
    from . import waymo as wm
    sb, ss, s_max, s_min = waymo_parse(sam3d_objs, zero_min_quirk=True)
    pb, ps, p_max, p_min = waymo_parse(pred_objs)
    pm, sm = match_samples(pb, sb, iou)
    best_alpha, best_score = 0, -1
    for alpha in waymo_alpha_grid(p_min, p_max, s_min, s_max):
        blob = wm.encode_objects(fuse_waymo(pb, ps, sb, ss, pm, sm, alpha))
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "wb") as f:
            f.write(blob)
        score = float(evaluate(out_path))
        if score > best_score:
            best_score, best_alpha = score, alpha
            with open(best_path, "wb") as f:
                f.write(blob)
        if verbose:
            print(f"Curr Score: {score},  Curr Alpha: {alpha}")
            print(f"Best Score: {best_score}, Best Alpha: {best_alpha}")
            print("-" * 80)
    return best_alpha, best_score
