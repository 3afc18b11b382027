"""Dataset-side I/O of the nuScenes entry point: the on-disk contract of the reference
(src/nuscenes/2d_to_3d.py:55-59,382-503,929-930 and gen_2d_masks_detic.py:497-506).

* `NuscTables` reads the nuScenes JSON tables directly (scene, sample, sample_data, ego_pose,
  calibrated_sensor, sensor, log); nuscenes-devkit is used when importable, but is not needed
  for the tables.
* lane centre-lines: `NuScenesMap.discretize_lanes(..., 0.5)` when the devkit is installed
  (2d_to_3d.py:228-240), otherwise a pre-discretised `<dataroot>/lanes/<location>.npy` (x,y,yaw).
* `frames_of_scene` assembles the per-frame kernel inputs (sweeps + transforms, camera records,
  RLE masks, labels) exactly as the reference gathers them at :422-503.
* `write_synthetic_dataset` lays a synthetic scene out in that same on-disk form, so the entry point
  can be exercised end to end without the real dataset.
"""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np

from . import geometry as geo
from . import synthetic as syn

CAM_LIST = ["CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_RIGHT", "CAM_BACK", "CAM_BACK_LEFT", "CAM_FRONT_LEFT"]
TABLES = ["scene", "sample", "sample_data", "ego_pose", "calibrated_sensor", "sensor", "log"]
ANNOTATION_TABLES = ["category", "attribute", "instance", "sample_annotation"]       # only the evaluation harness needs them


class NuscTables:
    def __init__(self, version, dataroot, annotations=True):
        """annotations=False: skip the ground-truth tables (the lifting path never reads them; on trainval they are the
        bulk of the JSON)."""
        self.version, self.dataroot = version, dataroot
        self.cam_templates = {}          # (calibrated_sensor token, ratio) -> camera record with the sensor's half filled in
        base = os.path.join(dataroot, version)
        self.t = {}
        for name in TABLES:
            with open(os.path.join(base, name + ".json")) as f:
                rows = json.load(f)
            self.t[name] = {r["token"]: r for r in rows}
        for name in ANNOTATION_TABLES if annotations else []:
            fn = os.path.join(base, name + ".json")
            if os.path.exists(fn):
                with open(fn) as f:
                    self.t[name] = {r["token"]: r for r in json.load(f)}
        # sample['anns'] like the devkit builds it
        self.anns_of = {}
        for ann in self.t.get("sample_annotation", {}).values():
            self.anns_of.setdefault(ann["sample_token"], []).append(ann["token"])
        sensors = self.t["sensor"]
        cs = self.t["calibrated_sensor"]
        # sample['data'][channel] like the devkit builds it: key frames only
        self.sample_data_of = {tok: {} for tok in self.t["sample"]}
        for sd in self.t["sample_data"].values():
            if sd.get("is_key_frame", False):
                ch = sensors[cs[sd["calibrated_sensor_token"]]["sensor_token"]]["channel"]
                self.sample_data_of[sd["sample_token"]][ch] = sd["token"]

    def get(self, table, token):
        return self.t[table][token]

    def scenes(self):
        return sorted(self.t["scene"].values(), key=lambda s: s["name"])

    def scene_by_name(self, name):
        for s in self.t["scene"].values():
            if s["name"] == name:
                return s
        raise KeyError(name)

    def samples_of_scene(self, scene):
        out, tok = [], scene["first_sample_token"]
        while tok != "":
            s = self.get("sample", tok)
            out.append(s)
            tok = s["next"]
        return out

    def category_name(self, ann):
        """the devkit's reverse-indexed `category_name` of a sample_annotation row"""
        return self.get("category", self.get("instance", ann["instance_token"])["category_token"])["name"]

    def location(self, scene):
        return self.get("log", scene["log_token"])["location"]


def load_lane_points(dataroot, location):
    """(L,3) float64 rows x, y, yaw of all lanes + lane connectors, discretised at 0.5 m.
    A pre-discretised `<dataroot>/lanes/<location>.npy` wins when present; otherwise nuscenes-devkit is used."""
    path = os.path.join(dataroot, "lanes", location + ".npy")
    if os.path.exists(path):
        return np.load(path).astype(np.float64).reshape(-1, 3)
    try:
        from nuscenes.map_expansion.map_api import NuScenesMap   # third-party, optional
    except ImportError:
        raise FileNotFoundError(f"nuscenes-devkit is not installed and {path} does not exist")
    nusc_map = NuScenesMap(dataroot=dataroot, map_name=location)
    records = nusc_map.lane + nusc_map.lane_connector
    poses = nusc_map.discretize_lanes([r["token"] for r in records], 0.5)
    pts = [p for lane in poses.values() for p in lane]
    return np.asarray(pts, np.float64).reshape(-1, 3)


def scene_manifest(tables: NuscTables, scene, mask_dir, n_sweeps=3, ratio=0.64, missing_ok=False):
    """What the frames of a scene consist of, without reading any bulk data (reference :415-503): per frame the sweep files
    with their transforms, the camera records, the mask file and the contents of the small <f>_data.json."""
    out = []
    for frame_num, sample in enumerate(tables.samples_of_scene(scene)):
        mp = os.path.join(mask_dir, scene["name"], f"{frame_num}_masks.pkl")
        dp = os.path.join(mask_dir, scene["name"], f"{frame_num}_data.json")
        if not (os.path.exists(mp) and os.path.exists(dp)):
            if missing_ok:       # the producer writes no files for frames without detections (gen_2d_masks_detic.py:490-491)
                mp, data = None, {"labels": [], "detection_scores": [], "cam_nums": []}
            else:
                raise FileNotFoundError(mp)
        else:
            with open(dp) as f:
                data = json.load(f)
        # sweeps: the key frame's LIDAR_TOP sample_data and its `next` chain (:433-463)
        sd = tables.get("sample_data", tables.sample_data_of[sample["token"]]["LIDAR_TOP"])
        key_pose = tables.get("ego_pose", sd["ego_pose_token"])
        paths, xfs = [], []
        for _ in range(n_sweeps):
            cs = tables.get("calibrated_sensor", sd["calibrated_sensor_token"])
            pose = tables.get("ego_pose", sd["ego_pose_token"])
            paths.append(os.path.join(tables.dataroot, sd["filename"]))
            xfs.append(geo.sweep_xf_record(cs["translation"], cs["rotation"], pose["translation"], pose["rotation"]))
            if sd.get("next", "") == "":
                break
            sd = tables.get("sample_data", sd["next"])
        cams = []
        for ch in CAM_LIST:
            csd = tables.get("sample_data", tables.sample_data_of[sample["token"]][ch])
            pose = tables.get("ego_pose", csd["ego_pose_token"])
            # a calibrated_sensor row serves every frame of its log: its half of the record (stage 1 and K') is built once
            key = (csd["calibrated_sensor_token"], ratio)
            tmpl = tables.cam_templates.get(key)
            if tmpl is None:
                cs = tables.get("calibrated_sensor", csd["calibrated_sensor_token"])
                # (rows that differ only in their token -- a table with one row per frame -- share the record as well)
                ckey = (tuple(cs["translation"]), tuple(cs["rotation"]), tuple(map(tuple, cs["camera_intrinsic"])), ratio)
                tmpl = tables.cam_templates.get(ckey)
                if tmpl is None:
                    tmpl = geo.nusc_cam_record([0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], cs["translation"], cs["rotation"], cs["camera_intrinsic"], ratio)
                    tables.cam_templates[ckey] = tmpl
                tables.cam_templates[key] = tmpl
            rec = tmpl.copy()
            rec[0:3] = (-np.asarray(pose["translation"], np.float64)).astype(np.float32)
            rec[3:12] = geo.quat_to_rotmat(pose["rotation"]).T.astype(np.float32).reshape(9)
            cams.append(rec)
        out.append(SimpleNamespace(token=sample["token"], sweep_paths=paths, sweep_xf=np.stack(xfs), cams=np.stack(cams), mask_path=mp,
                                   labels=list(data["labels"]), scores=list(data["detection_scores"]), cam_nums=list(data["cam_nums"]),
                                   ego_xyz=np.asarray(key_pose["translation"], np.float64), ratio=ratio))
    return out


def frames_of_scene(tables: NuscTables, scene, mask_dir, n_sweeps=3, ratio=0.64, missing_ok=False):
    """Kernel inputs of every frame of a scene (reference :415-503), read frame by frame like the reference does:
    pickle.load of the mask file (:422-424), np.fromfile of every sweep (utils/pcd.py:250)."""
    frames = []
    for m in scene_manifest(tables, scene, mask_dir, n_sweeps, ratio, missing_ok):
        rles = []
        if m.mask_path is not None:
            with open(m.mask_path, "rb") as f:
                rles = pickle.load(f)
        raws = [np.fromfile(p, dtype=np.float32).reshape(-1, 5) for p in m.sweep_paths]
        if rles:
            W, H = rles[0]["size"]
        else:
            W, H = int(1600 * ratio), int(900 * ratio)
        frames.append(SimpleNamespace(
            token=m.token, sweeps_raw=raws, sweep_xf=m.sweep_xf, cams=m.cams, rles=list(rles), labels=m.labels, scores=m.scores,
            cam_nums=m.cam_nums, ego_xyz=m.ego_xyz, width=int(W), height=int(H)))
    return frames


# --------------------------------------------------------------------------- synthetic dataset on disk
def write_synthetic_dataset(root, cfg: syn.SyntheticConfig, n_scenes=2, frames_per_scene=4, version="v1.0-synth",
                            mask_subdir="mask_outputs/nuscenes-detic", lane_points=4000, pool=0):
    """Writes `<root>/data/nuScenes/{<version>/*.json, sweeps/..., lanes/<loc>.npy}` and
    `<root>/<mask_subdir>/<scene>/<f>_{masks.pkl,data.json}` from synthetic frames.
    pool > 0: only that many distinct frames are generated and reused in turn (large datasets for throughput runs).
    Returns (dataroot, mask_dir, scene_names)."""
    made = {}

    def frame_of(idx):
        key = idx % pool if pool > 0 else idx
        if key not in made:
            made[key] = syn.make_frame(cfg, key)
        return made[key]
    dataroot = os.path.join(root, "data", "nuScenes")
    mask_dir = os.path.join(root, mask_subdir)
    os.makedirs(os.path.join(dataroot, version), exist_ok=True)
    os.makedirs(os.path.join(dataroot, "sweeps", "LIDAR_TOP"), exist_ok=True)
    os.makedirs(os.path.join(dataroot, "lanes"), exist_ok=True)
    tabs = {k: [] for k in TABLES + ANNOTATION_TABLES}
    tok = lambda kind, *a: f"{kind}-" + "-".join(str(x) for x in a)
    # ground truth of the synthetic objects (evaluation harness): one category per detection class
    category_of = {"car": "vehicle.car", "truck": "vehicle.truck", "bus": "vehicle.bus.rigid", "trailer": "vehicle.trailer",
                   "construction_vehicle": "vehicle.construction", "pedestrian": "human.pedestrian.adult",
                   "motorcycle": "vehicle.motorcycle", "bicycle": "vehicle.bicycle", "traffic_cone": "movable_object.trafficcone",
                   "barrier": "movable_object.barrier"}
    for cat in sorted(set(category_of.values())):
        tabs["category"].append({"token": tok("cat", cat), "name": cat})
    from .lifting import get_detection_name
    for ch in ["LIDAR_TOP"] + CAM_LIST:
        tabs["sensor"].append({"token": tok("sensor", ch), "channel": ch, "modality": "lidar" if ch == "LIDAR_TOP" else "camera"})
    names = []
    for s in range(n_scenes):
        name = f"scene-{9000 + s:04d}"
        names.append(name)
        loc = f"synthtown-{s}"
        tabs["log"].append({"token": tok("log", s), "location": loc})
        os.makedirs(os.path.join(mask_dir, name), exist_ok=True)
        first_center = None
        sample_tokens = [tok("sample", s, f) for f in range(frames_per_scene)]
        for f in range(frames_per_scene):
            fr = frame_of(s * 1000 + f)
            if first_center is None:
                first_center = fr.ego_xyz[:2].copy()
            st = sample_tokens[f]
            tabs["sample"].append({"token": st, "scene_token": tok("scene", s), "timestamp": 500000 * f,
                                   "prev": sample_tokens[f - 1] if f else "", "next": sample_tokens[f + 1] if f + 1 < frames_per_scene else ""})
            # lidar sweeps: a `next`-linked chain, the first one is the key frame
            for k, (raw, xf) in enumerate(zip(fr.sweeps_raw, fr.sweep_xf)):
                fn = os.path.join("sweeps", "LIDAR_TOP", f"{name}_{f}_{k}.bin")
                raw.astype(np.float32).tofile(os.path.join(dataroot, fn))
                cst, ept = tok("cs", s, f, "L", k), tok("pose", s, f, "L", k)
                Rcs, tcs = xf[0:9].reshape(3, 3).astype(np.float64), xf[9:12].astype(np.float64)
                Reg, teg = xf[12:21].reshape(3, 3).astype(np.float64), xf[21:24].astype(np.float64)
                tabs["calibrated_sensor"].append({"token": cst, "sensor_token": tok("sensor", "LIDAR_TOP"),
                                                  "translation": tcs.tolist(), "rotation": geo.rotmat_to_quat(Rcs).tolist(), "camera_intrinsic": []})
                tabs["ego_pose"].append({"token": ept, "translation": teg.tolist() if k else fr.ego_xyz.tolist(),
                                         "rotation": geo.rotmat_to_quat(Reg).tolist()})
                tabs["sample_data"].append({"token": tok("sd", s, f, "L", k), "sample_token": st, "ego_pose_token": ept,
                                            "calibrated_sensor_token": cst, "filename": fn, "is_key_frame": k == 0,
                                            "next": tok("sd", s, f, "L", k + 1) if k + 1 < len(fr.sweeps_raw) else "", "prev": ""})
            for c, ch in enumerate(CAM_LIST[: fr.cams.shape[0]]):
                cst, ept = tok("cs", s, f, ch), tok("pose", s, f, ch)
                # invert the record: stage 0 = (-t_ego, R_ego^T), stage 1 = (-t_cs, R_cs^T), K' = K*ratio
                t_ego_neg, R_egoT, _ = geo.cam_stage(fr.cams[c], 0)
                t_cs_neg, R_csT, _ = geo.cam_stage(fr.cams[c], 1)
                K = geo.cam_K(fr.cams[c]) / cfg.ratio
                K[2, 2] = 1.0
                tabs["calibrated_sensor"].append({"token": cst, "sensor_token": tok("sensor", ch), "translation": (-t_cs_neg).tolist(),
                                                  "rotation": geo.rotmat_to_quat(R_csT.T).tolist(), "camera_intrinsic": K.tolist()})
                tabs["ego_pose"].append({"token": ept, "translation": (-t_ego_neg).tolist(),
                                         "rotation": geo.rotmat_to_quat(R_egoT.T).tolist()})
                tabs["sample_data"].append({"token": tok("sd", s, f, ch), "sample_token": st, "ego_pose_token": ept,
                                            "calibrated_sensor_token": cst, "filename": f"samples/{ch}/{name}_{f}.jpg",
                                            "is_key_frame": True, "next": "", "prev": ""})
            for j, obj in enumerate(fr.meta.get("objects", [])):
                det = get_detection_name(obj["label"])
                if det not in category_of:
                    continue
                itok = tok("inst", s, f, j)          # synthetic objects are redrawn every frame: one instance per annotation
                tabs["instance"].append({"token": itok, "category_token": tok("cat", category_of[det]), "nbr_annotations": 1,
                                         "first_annotation_token": tok("ann", s, f, j), "last_annotation_token": tok("ann", s, f, j)})
                tabs["sample_annotation"].append({"token": tok("ann", s, f, j), "sample_token": st, "instance_token": itok,
                                                  "attribute_tokens": [], "translation": obj["center"], "size": obj["size"],
                                                  "rotation": [1.0, 0.0, 0.0, 0.0], "prev": "", "next": "",
                                                  "num_lidar_pts": 10, "num_radar_pts": 0, "visibility_token": "4"})
            with open(os.path.join(mask_dir, name, f"{f}_masks.pkl"), "wb") as fh:
                pickle.dump(fr.rles, fh)
            with open(os.path.join(mask_dir, name, f"{f}_data.json"), "w") as fh:
                json.dump({"labels": fr.labels, "detection_scores": fr.scores, "cam_nums": fr.cam_nums}, fh)
        tabs["scene"].append({"token": tok("scene", s), "name": name, "log_token": tok("log", s),
                              "first_sample_token": sample_tokens[0], "nbr_samples": frames_per_scene})
        np.save(os.path.join(dataroot, "lanes", loc + ".npy"), syn.make_lane_table(first_center, lane_points, seed=100 + s))
        # drivable area in the map-expansion layout (node / polygon / drivable_area tables): a 160 m square around the first
        # ego position with a 24 m square hole to its north-east -- what eval_custom's drivable filter reads (:496-505)
        os.makedirs(os.path.join(dataroot, "maps", "expansion"), exist_ok=True)
        cx, cy = float(first_center[0]), float(first_center[1])
        rings = {"ext": [(cx - 80, cy - 80), (cx + 80, cy - 80), (cx + 80, cy + 80), (cx - 80, cy + 80)],
                 "hole": [(cx + 20, cy + 20), (cx + 44, cy + 20), (cx + 44, cy + 44), (cx + 20, cy + 44)]}
        nodes = [{"token": f"node-{s}-{k}-{i}", "x": x, "y": y} for k, pts in rings.items() for i, (x, y) in enumerate(pts)]
        polygon = {"token": f"poly-{s}", "exterior_node_tokens": [f"node-{s}-ext-{i}" for i in range(4)],
                   "holes": [{"node_tokens": [f"node-{s}-hole-{i}" for i in range(4)]}]}
        with open(os.path.join(dataroot, "maps", "expansion", loc + ".json"), "w") as fh:
            json.dump({"node": nodes, "polygon": [polygon], "drivable_area": [{"token": f"da-{s}", "polygon_tokens": [polygon["token"]]}]}, fh)
    for k, rows in tabs.items():
        with open(os.path.join(dataroot, version, k + ".json"), "w") as fh:
            json.dump(rows, fh)
    return dataroot, mask_dir, names
