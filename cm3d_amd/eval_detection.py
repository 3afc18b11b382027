"""nuScenes-style detection evaluation of pseudo-labels.

Host-side restatement of the reference's `src/nuscenes/eval_custom.py` (file:line cites below are into that file)
on top of this package's own table reader (`nusc_io.NuscTables`) -- nuscenes-devkit, shapely and pyquaternion are
not needed.  Everything is numpy; the work is tiny (a few boxes per sample) and runs once per result file, so
there is no kernel here.

What is mirrored
  * `DetectionConfig` (:36-100) and the CVPR-2019 defaults of nuscenes-devkit's `detection_cvpr_2019.json`;
  * `load_gt` (:296-404) incl. `category_to_detection_name[_rare]` (:204-262) and the devkit's `box_velocity`;
  * `add_center_dist` (:103-127), `filter_eval_boxes` (:441-536: range, zero-point, bike-rack and drivable-area filters; the
    drivable polygons are read from the map-expansion json directly, `load_drivable_polygons`);
  * `accumulate_object_class` (:542-706, class-agnostic "object" matching) and `accumulate_with_recall` (:709-864);
  * `calc_ap` / `calc_tp` and the `DetectionMetrics` summary (mAP, TP errors, NDS) of nuscenes-devkit 1.1.10;
  * `DetectionEval.evaluate/main` (:866-1155): `metrics_summary.json`, `metrics_details.json`, the printed table.

Parity: the reference module imports nuscenes-devkit, pyquaternion and shapely at module level, none of which is installed
here.  tests/golden/gen_golden_eval.py imports it all the same -- with those third-party helpers restated underneath -- and
freezes what the reference's OWN functions (`load_gt`, `add_center_dist`, `filter_eval_boxes`, `accumulate_object_class`,
`accumulate_with_recall`) return on a synthetic table set (golden G10); tests/test_eval_detection.py holds this module to
it box for box and curve for curve (1e-12).  What stays unpinned is the third-party helper behaviour itself (devkit
`box_velocity`, `points_in_box`, shapely `within`, ...), restated here and in the generator from published behaviour;
hand-computed known answers cover those.
"""
import json
import os
import time

import numpy as np

from . import geometry as geo

DETECTION_NAMES = ['car', 'truck', 'bus', 'trailer', 'construction_vehicle', 'pedestrian', 'motorcycle', 'bicycle',
                   'traffic_cone', 'barrier']
TP_METRICS = ['trans_err', 'scale_err', 'orient_err', 'vel_err', 'attr_err']
NELEM = 101                                  # DetectionMetricData.nelem

# nuscenes-devkit detection_cvpr_2019.json
CVPR_2019 = {
    "class_range": {"car": 50, "truck": 50, "bus": 50, "trailer": 50, "construction_vehicle": 50, "pedestrian": 40,
                    "motorcycle": 40, "bicycle": 40, "traffic_cone": 30, "barrier": 30},
    "dist_fcn": "center_distance", "dist_ths": [0.5, 1.0, 2.0, 4.0], "dist_th_tp": 2.0, "min_recall": 0.1,
    "min_precision": 0.1, "max_boxes_per_sample": 500, "mean_ap_weight": 5,
}

_DETECTION_MAPPING = {            # devkit category_to_detection_name
    'movable_object.barrier': 'barrier', 'vehicle.bicycle': 'bicycle', 'vehicle.bus.bendy': 'bus', 'vehicle.bus.rigid': 'bus',
    'vehicle.car': 'car', 'vehicle.construction': 'construction_vehicle', 'vehicle.motorcycle': 'motorcycle',
    'human.pedestrian.adult': 'pedestrian', 'human.pedestrian.child': 'pedestrian',
    'human.pedestrian.construction_worker': 'pedestrian', 'human.pedestrian.police_officer': 'pedestrian',
    'movable_object.trafficcone': 'traffic_cone', 'vehicle.trailer': 'trailer', 'vehicle.truck': 'truck',
}
_DETECTION_MAPPING_RARE = dict(_DETECTION_MAPPING, **{'human.pedestrian.child': 'child', 'human.pedestrian.stroller': 'stroller'})  # :204-232


def category_to_detection_name(category_name, rare=False):
    return (_DETECTION_MAPPING_RARE if rare else _DETECTION_MAPPING).get(category_name)


class DetectionConfig:
    """:36-100"""

    def __init__(self, class_range, dist_fcn, dist_ths, dist_th_tp, min_recall, min_precision, max_boxes_per_sample, mean_ap_weight):
        assert dist_th_tp in dist_ths, "dist_th_tp must be in set of dist_ths."
        if dist_fcn != 'center_distance':
            raise Exception('Error: Unknown distance function %s!' % dist_fcn)
        self.class_range, self.dist_fcn, self.dist_ths, self.dist_th_tp = class_range, dist_fcn, dist_ths, dist_th_tp
        self.min_recall, self.min_precision = min_recall, min_precision
        self.max_boxes_per_sample, self.mean_ap_weight = max_boxes_per_sample, mean_ap_weight
        self.class_names = list(class_range.keys())

    def serialize(self):
        return {k: getattr(self, k) for k in ('class_range', 'dist_fcn', 'dist_ths', 'dist_th_tp', 'min_recall', 'min_precision',
                                              'max_boxes_per_sample', 'mean_ap_weight')}

    @classmethod
    def deserialize(cls, c):
        return cls(c['class_range'], c['dist_fcn'], c['dist_ths'], c['dist_th_tp'], c['min_recall'], c['min_precision'],
                   c['max_boxes_per_sample'], c['mean_ap_weight'])


def config_factory(name='detection_cvpr_2019'):
    if name != 'detection_cvpr_2019':
        raise ValueError(name)
    return DetectionConfig.deserialize(json.loads(json.dumps(CVPR_2019)))


class EvalBoxes:
    """sample_token -> list of box dicts (translation, size, rotation, velocity, num_pts, detection_name,
    detection_score, attribute_name, ego_translation)."""

    def __init__(self):
        self.boxes = {}

    def __getitem__(self, tok):
        return self.boxes[tok]

    def add_boxes(self, tok, boxes):
        self.boxes[tok] = list(boxes)

    @property
    def sample_tokens(self):
        return list(self.boxes.keys())

    @property
    def all(self):
        return [b for tok in self.sample_tokens for b in self.boxes[tok]]


def _box(sample_token, translation, size, rotation, velocity=(0.0, 0.0), num_pts=-1, detection_name='car', detection_score=-1.0,
         attribute_name=''):
    return dict(sample_token=sample_token, translation=tuple(float(v) for v in translation), size=tuple(float(v) for v in size),
                rotation=tuple(float(v) for v in rotation), velocity=tuple(float(v) for v in velocity), num_pts=int(num_pts),
                detection_name=detection_name, detection_score=float(detection_score), attribute_name=attribute_name,
                ego_translation=(0.0, 0.0, 0.0))


def ego_dist(box):
    return float(np.sqrt(np.sum(np.array(box['ego_translation'][:2]) ** 2)))


# ----------------------------------------------------------------------------- loaders
def load_prediction(result_path, max_boxes_per_sample, verbose=False):
    """devkit loaders.load_prediction: the submission JSON -> EvalBoxes, meta."""
    with open(result_path) as f:
        data = json.load(f)
    assert 'results' in data, 'Error: No field `results` in result file.'
    out = EvalBoxes()
    for tok, boxes in data['results'].items():
        out.add_boxes(tok, [_box(tok, b['translation'], b['size'], b['rotation'], b.get('velocity', (0, 0)), -1, b['detection_name'],
                                 b['detection_score'], b.get('attribute_name', '')) for b in boxes])
        assert len(out[tok]) <= max_boxes_per_sample, "Error: Only <= %d boxes per sample allowed!" % max_boxes_per_sample
    if verbose:
        print("Loaded results from {}. Found detections for {} samples.".format(result_path, len(out.sample_tokens)))
    return out, data.get('meta', {})


def box_velocity(tables, ann, max_time_diff=1.5):
    """devkit NuScenes.box_velocity: finite difference over the instance's previous / next annotation."""
    has_prev, has_next = ann['prev'] != '', ann['next'] != ''
    if not has_prev and not has_next:
        return np.array([np.nan, np.nan, np.nan])
    first = tables.get('sample_annotation', ann['prev']) if has_prev else ann
    last = tables.get('sample_annotation', ann['next']) if has_next else ann
    t_first = 1e-6 * tables.get('sample', first['sample_token'])['timestamp']
    t_last = 1e-6 * tables.get('sample', last['sample_token'])['timestamp']
    dt = t_last - t_first
    if has_next and has_prev:
        max_time_diff *= 2          # two-sample difference
    if dt > max_time_diff or dt <= 0:
        return np.array([np.nan, np.nan, np.nan])
    return (np.array(last['translation']) - np.array(first['translation'])) / dt


def load_gt(tables, scene_names=None, verbose=False, rare=False):
    """:296-404.  `scene_names` plays the role of the split (None = every scene of the tables)."""
    attribute_map = {a['token']: a['name'] for a in tables.t.get('attribute', {}).values()}
    out = EvalBoxes()
    for sample in tables.t['sample'].values():
        scene = tables.get('scene', sample['scene_token'])
        if scene_names is not None and scene['name'] not in scene_names:
            continue
        boxes = []
        for tok in tables.anns_of.get(sample['token'], []):
            ann = tables.get('sample_annotation', tok)
            name = category_to_detection_name(tables.category_name(ann), rare)
            if name is None:
                continue
            attrs = ann.get('attribute_tokens', [])
            if len(attrs) > 1:
                raise Exception('Error: GT annotations must not have more than one attribute!')
            boxes.append(_box(sample['token'], ann['translation'], ann['size'], ann['rotation'], box_velocity(tables, ann)[:2],
                              ann['num_lidar_pts'] + ann['num_radar_pts'], name, -1.0, attribute_map[attrs[0]] if attrs else ''))
        out.add_boxes(sample['token'], boxes)
    if verbose:
        print("Loaded ground truth annotations for {} samples.".format(len(out.sample_tokens)))
    return out


def add_center_dist(tables, eval_boxes):
    """:103-127"""
    for tok in eval_boxes.sample_tokens:
        sd = tables.get('sample_data', tables.sample_data_of[tok]['LIDAR_TOP'])
        pose = tables.get('ego_pose', sd['ego_pose_token'])
        for b in eval_boxes[tok]:
            b['ego_translation'] = tuple(b['translation'][i] - pose['translation'][i] for i in range(3))
    return eval_boxes


def _point_in_box(center, size_wlh, rotation, p):
    """devkit geometry_utils.points_in_box for one point: inside the oriented box (w along y, l along x)."""
    R = geo.quat_to_rotmat(rotation)
    local = R.T @ (np.asarray(p, float) - np.asarray(center, float))
    w, l, h = size_wlh
    return abs(local[0]) <= l / 2 and abs(local[1]) <= w / 2 and abs(local[2]) <= h / 2


def load_drivable_polygons(dataroot, location):
    """The drivable-area polygons of a map, read straight from the nuScenes map-expansion file
    `<dataroot>/maps/expansion/<location>.json` (what NuScenesMap.drivable_area + extract_polygon give at :496-505): every
    polygon token of every drivable_area record -> (exterior ring (n,2), [hole rings]).  No devkit, no shapely."""
    path = os.path.join(dataroot, "maps", "expansion", location + ".json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"drivable-area filtering needs {path}")
    with open(path) as f:
        m = json.load(f)
    node = {n["token"]: (float(n["x"]), float(n["y"])) for n in m["node"]}
    poly = {p["token"]: p for p in m["polygon"]}
    out = []
    for rec in m["drivable_area"]:
        for tok in rec["polygon_tokens"]:
            p = poly[tok]
            ext = np.array([node[t] for t in p["exterior_node_tokens"]], np.float64).reshape(-1, 2)
            holes = [np.array([node[t] for t in h["node_tokens"]], np.float64).reshape(-1, 2) for h in p.get("holes", [])]
            out.append((ext, holes))
    return out


def _inside_ring(ring, x, y):
    """Even-odd rule (strictly inside for points off the boundary, like shapely's `within` up to a set of measure zero)."""
    xs, ys = ring[:, 0], ring[:, 1]
    xn, yn = np.roll(xs, -1), np.roll(ys, -1)
    crosses = ((ys > y) != (yn > y)) & (x < (xn - xs) * (y - ys) / np.where(yn == ys, 1.0, yn - ys) + xs)
    return bool(np.count_nonzero(crosses) & 1)


def point_in_polygons(polygons, x, y):
    """`any(Point(x, y).within(polygon))` of :514-519: inside an exterior ring and in none of its holes."""
    for ext, holes in polygons:
        if ext.shape[0] >= 3 and _inside_ring(ext, x, y) and not any(h.shape[0] >= 3 and _inside_ring(h, x, y) for h in holes):
            return True
    return False


def filter_eval_boxes(tables, eval_boxes, max_dist, drivable_filtering=False, verbose=False):
    """:441-536: distance, zero points, bike racks, and -- when asked for -- the drivable-area filter (:489-526): a box stays
    only if its centre lies within a drivable-area polygon of the map of the FIRST sample's scene (like the reference, which
    looks the map up once, :491-493)."""
    total = dist_f = point_f = rack_f = 0
    for tok in eval_boxes.sample_tokens:
        total += len(eval_boxes[tok])
        eval_boxes.boxes[tok] = [b for b in eval_boxes[tok] if ego_dist(b) < max_dist[b['detection_name']]]
        dist_f += len(eval_boxes[tok])
        eval_boxes.boxes[tok] = [b for b in eval_boxes[tok] if not b['num_pts'] == 0]
        point_f += len(eval_boxes[tok])
        racks = [tables.get('sample_annotation', a) for a in tables.anns_of.get(tok, [])
                 if tables.category_name(tables.get('sample_annotation', a)) == 'static_object.bicycle_rack']
        kept = []
        for b in eval_boxes[tok]:
            if b['detection_name'] in ['bicycle', 'motorcycle'] and any(
                    _point_in_box(r['translation'], r['size'], r['rotation'], b['translation']) for r in racks):
                continue
            kept.append(b)
        eval_boxes.boxes[tok] = kept
        rack_f += len(kept)
    if verbose:
        print("> Original number of boxes: %d" % total)
        print("> After distance based filtering: %d" % dist_f)
        print("> After LIDAR and RADAR points based filtering: %d" % point_f)
        print("> After bike rack filtering: %d" % rack_f)
    if drivable_filtering and eval_boxes.sample_tokens:
        first = tables.get('sample', eval_boxes.sample_tokens[0])
        scene = tables.get('scene', first['scene_token'])
        polygons = load_drivable_polygons(tables.dataroot, tables.location(scene))
        driv_f = 0
        for tok in eval_boxes.sample_tokens:
            eval_boxes.boxes[tok] = [b for b in eval_boxes[tok] if point_in_polygons(polygons, b['translation'][0], b['translation'][1])]
            driv_f += len(eval_boxes[tok])
        if verbose:
            print("> After drivable area filtering: %d" % driv_f)
    return eval_boxes


# ----------------------------------------------------------------------------- per-match measures (devkit eval/common/utils.py)
def center_distance(gt, pred):
    return float(np.linalg.norm(np.array(pred['translation'][:2]) - np.array(gt['translation'][:2])))


def velocity_l2(gt, pred):
    return float(np.linalg.norm(np.array(pred['velocity']) - np.array(gt['velocity'])))


def quaternion_yaw(q):
    v = geo.quat_to_rotmat(q) @ np.array([1.0, 0.0, 0.0])
    return float(np.arctan2(v[1], v[0]))


def angle_diff(x, y, period):
    diff = (x - y + period / 2) % period - period / 2
    if diff > np.pi:
        diff = diff - (2 * np.pi)
    return diff


def yaw_diff(gt, pred, period=2 * np.pi):
    return abs(angle_diff(quaternion_yaw(gt['rotation']), quaternion_yaw(pred['rotation']), period))


def scale_iou(a, b):
    sa, sb = np.array(a['size']), np.array(b['size'])
    assert all(sa > 0) and all(sb > 0), 'Error: box sizes must be >0.'
    inter = np.prod(np.minimum(sa, sb))
    return float(inter / (np.prod(sa) + np.prod(sb) - inter))


def attr_acc(gt, pred):
    if gt['attribute_name'] == '':
        return np.nan              # GT without attribute: excluded from the mean
    return float(gt['attribute_name'] == pred['attribute_name'])


def cummean(x):
    if sum(np.isnan(x)) == len(x):
        return np.ones(len(x))
    sum_vals = np.nancumsum(x.astype(float))
    count_vals = np.cumsum(~np.isnan(x))
    return np.divide(sum_vals, count_vals, out=np.zeros_like(sum_vals), where=count_vals != 0)


class DetectionMetricData:
    nelem = NELEM

    def __init__(self, recall, precision, confidence, trans_err, vel_err, scale_err, orient_err, attr_err):
        self.recall, self.precision, self.confidence = recall, precision, confidence
        self.trans_err, self.vel_err, self.scale_err, self.orient_err, self.attr_err = trans_err, vel_err, scale_err, orient_err, attr_err

    @classmethod
    def no_predictions(cls):
        return cls(np.linspace(0, 1, NELEM), np.zeros(NELEM), np.zeros(NELEM), np.ones(NELEM), np.ones(NELEM), np.ones(NELEM),
                   np.ones(NELEM), np.ones(NELEM))

    @property
    def max_recall_ind(self):
        non_zero = np.nonzero(self.confidence)[0]
        return 0 if len(non_zero) == 0 else int(non_zero[-1])

    def serialize(self):
        return {k: np.asarray(getattr(self, k)).tolist() for k in ('recall', 'precision', 'confidence', 'trans_err', 'vel_err',
                                                                   'scale_err', 'orient_err', 'attr_err')}


def _accumulate(gt_boxes, pred_boxes, class_name, dist_th, object_class):
    """The common body of accumulate_object_class (:542-706, class_name ignored) and accumulate_with_recall (:709-864).
    Returns (DetectionMetricData, actual recall)."""
    gt_all = gt_boxes.all
    npos = len(gt_all) if object_class else len([1 for g in gt_all if g['detection_name'] == class_name])
    if npos == 0:
        return DetectionMetricData.no_predictions(), 0
    preds = [b for b in pred_boxes.all if object_class or b['detection_name'] == class_name]
    confs = [b['detection_score'] for b in preds]
    sortind = [i for (v, i) in sorted((v, i) for (i, v) in enumerate(confs))][::-1]
    tp, fp, conf = [], [], []
    match = {k: [] for k in ('trans_err', 'vel_err', 'scale_err', 'orient_err', 'attr_err', 'conf')}
    taken = set()
    for ind in sortind:
        p = preds[ind]
        min_dist, match_idx = np.inf, None
        for gi, g in enumerate(gt_boxes.boxes.get(p['sample_token'], [])):
            if (object_class or g['detection_name'] == class_name) and (p['sample_token'], gi) not in taken:
                d = center_distance(g, p)
                if d < min_dist:
                    min_dist, match_idx = d, gi
        if min_dist < dist_th:
            taken.add((p['sample_token'], match_idx))
            tp.append(1); fp.append(0); conf.append(p['detection_score'])
            g = gt_boxes[p['sample_token']][match_idx]
            match['trans_err'].append(center_distance(g, p))
            if object_class:
                # :629-650 -- per matched GT class, with the half-circle period for everything
                match['vel_err'].append(np.nan if g['detection_name'] in ['traffic_cone', 'barrier'] else velocity_l2(g, p))
                match['scale_err'].append(1 - scale_iou(g, p))
                match['orient_err'].append(np.nan if g['detection_name'] in ['traffic_cone'] else yaw_diff(g, p, period=np.pi))
                match['attr_err'].append(np.nan if g['detection_name'] in ['barrier', 'traffic_cone'] else 1 - attr_acc(g, p))
            else:
                match['vel_err'].append(velocity_l2(g, p))
                match['scale_err'].append(1 - scale_iou(g, p))
                match['orient_err'].append(yaw_diff(g, p, period=np.pi if class_name == 'barrier' else 2 * np.pi))
                match['attr_err'].append(1 - attr_acc(g, p))
            match['conf'].append(p['detection_score'])
        else:
            tp.append(0); fp.append(1); conf.append(p['detection_score'])
    if len(match['trans_err']) == 0:
        return DetectionMetricData.no_predictions(), 0
    tp = np.cumsum(tp).astype(float)
    fp = np.cumsum(fp).astype(float)
    conf = np.array(conf)
    prec = tp / (fp + tp)
    rec = tp / float(npos)
    rec_actual = float(np.max(rec))
    rec_interp = np.linspace(0, 1, NELEM)
    prec = np.interp(rec_interp, rec, prec, right=0)
    conf = np.interp(rec_interp, rec, conf, right=0)
    for key in match:
        if key == 'conf':
            continue
        tmp = cummean(np.array(match[key], dtype=float))
        match[key] = np.interp(conf[::-1], match['conf'][::-1], tmp[::-1])[::-1]
    return DetectionMetricData(rec_interp, prec, conf, match['trans_err'], match['vel_err'], match['scale_err'], match['orient_err'],
                               match['attr_err']), rec_actual


def accumulate_object_class(gt_boxes, pred_boxes, dist_fcn=None, dist_th=2.0, verbose=False):
    """:542-706 -> (metric data, actual recall)"""
    return _accumulate(gt_boxes, pred_boxes, "object", dist_th, True)


def accumulate_with_recall(gt_boxes, pred_boxes, class_name, dist_fcn=None, dist_th=2.0, verbose=False):
    """:709-864 -> (actual recall, metric data)"""
    md, rec = _accumulate(gt_boxes, pred_boxes, class_name, dist_th, False)
    return rec, md


def calc_ap(md, min_recall, min_precision):
    """devkit algo.calc_ap"""
    assert 0 <= min_precision < 1 and 0 <= min_recall <= 1
    prec = np.copy(md.precision)
    prec = prec[round(100 * min_recall) + 1:]
    prec -= min_precision
    prec[prec < 0] = 0
    return float(np.mean(prec)) / (1.0 - min_precision)


def calc_tp(md, min_recall, metric_name):
    """devkit algo.calc_tp"""
    first_ind = round(100 * min_recall) + 1
    last_ind = md.max_recall_ind
    if last_ind < first_ind:
        return 1.0
    return float(np.mean(getattr(md, metric_name)[first_ind:last_ind + 1]))


class DetectionMetrics:
    """devkit eval/detection/data_classes.DetectionMetrics"""

    def __init__(self, cfg):
        self.cfg = cfg
        self._label_aps, self._label_tp_errors = {}, {}
        self.eval_time = None

    def add_label_ap(self, name, dist_th, ap):
        self._label_aps.setdefault(name, {})[dist_th] = ap

    def add_label_tp(self, name, metric, tp):
        self._label_tp_errors.setdefault(name, {})[metric] = tp

    def add_runtime(self, t):
        self.eval_time = t

    @property
    def mean_dist_aps(self):
        return {n: float(np.mean(list(d.values()))) for n, d in self._label_aps.items()}

    @property
    def mean_ap(self):
        return float(np.mean(list(self.mean_dist_aps.values())))

    @property
    def tp_errors(self):
        return {m: float(np.nanmean([self._label_tp_errors[n][m] for n in self._label_tp_errors])) for m in TP_METRICS}

    @property
    def tp_scores(self):
        out = {}
        for m, err in self.tp_errors.items():
            out[m] = max(1.0 - err, 0.0)
        return out

    @property
    def nd_score(self):
        total = float(self.cfg.mean_ap_weight * self.mean_ap + np.sum(list(self.tp_scores.values())))
        return total / float(self.cfg.mean_ap_weight + len(self.tp_scores))

    def serialize(self):
        return {'label_aps': {n: {str(k): v for k, v in d.items()} for n, d in self._label_aps.items()},
                'mean_dist_aps': self.mean_dist_aps, 'mean_ap': self.mean_ap, 'label_tp_errors': self._label_tp_errors,
                'tp_errors': self.tp_errors, 'tp_scores': self.tp_scores, 'nd_score': self.nd_score, 'eval_time': self.eval_time,
                'cfg': self.cfg.serialize()}


class DetectionEval:
    """:866-1155"""

    def __init__(self, tables, config, result_path, eval_set=None, output_dir=None, drivable_filtering=False, object_only=True,
                 verbose=True):
        self.tables, self.cfg, self.result_path, self.eval_set = tables, config, result_path, eval_set
        self.output_dir, self.verbose, self.object_only = output_dir, verbose, object_only
        assert os.path.exists(result_path), 'Error: The result file does not exist!'
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
        self.pred_boxes, self.meta = load_prediction(result_path, config.max_boxes_per_sample, verbose=verbose)
        self.gt_boxes = load_gt(tables, eval_set, verbose=verbose, rare=len(config.class_range) > 10)
        self.pred_boxes = add_center_dist(tables, self.pred_boxes)
        self.gt_boxes = add_center_dist(tables, self.gt_boxes)
        if verbose:
            print('Filtering predictions')
        self.pred_boxes = filter_eval_boxes(tables, self.pred_boxes, config.class_range, drivable_filtering, verbose)
        if verbose:
            print('Filtering ground truth annotations')
        self.gt_boxes = filter_eval_boxes(tables, self.gt_boxes, config.class_range, drivable_filtering, verbose)
        self.sample_tokens = self.gt_boxes.sample_tokens

    def evaluate(self):
        t0 = time.time()
        recall_list, md_list = [], {}
        names = ["object"] if self.object_only else list(self.cfg.class_names)
        for name in names:
            recs = []
            for th in self.cfg.dist_ths:
                if self.object_only:
                    md, rec = accumulate_object_class(self.gt_boxes, self.pred_boxes, None, th)
                else:
                    rec, md = accumulate_with_recall(self.gt_boxes, self.pred_boxes, name, None, th)
                md_list[(name, th)] = md
                recs.append(rec)
            recall_list.append(sum(recs) / len(recs))
        metrics = DetectionMetrics(self.cfg)
        for name in names:
            for th in self.cfg.dist_ths:
                metrics.add_label_ap(name, th, calc_ap(md_list[(name, th)], self.cfg.min_recall, self.cfg.min_precision))
            for m in TP_METRICS:
                md = md_list[(name, self.cfg.dist_th_tp)]
                if not self.object_only and name in ['traffic_cone'] and m in ['attr_err', 'vel_err', 'orient_err']:
                    tp = np.nan
                elif not self.object_only and name in ['barrier'] and m in ['attr_err', 'vel_err']:
                    tp = np.nan
                else:
                    tp = calc_tp(md, self.cfg.min_recall, m)
                metrics.add_label_tp(name, m, tp)
        metrics.add_runtime(time.time() - t0)
        return metrics, md_list, recall_list

    def main(self):
        metrics, md_list, recall_list = self.evaluate()
        summary = metrics.serialize()
        summary['meta'] = dict(self.meta)
        summary['mean_recall'] = sum(recall_list) / len(recall_list)
        if self.output_dir:
            with open(os.path.join(self.output_dir, 'metrics_summary.json'), 'w') as f:
                json.dump(summary, f, indent=2)
            with open(os.path.join(self.output_dir, 'metrics_details.json'), 'w') as f:
                json.dump({"%s:%s" % k: v.serialize() for k, v in md_list.items()}, f, indent=2)
        if self.verbose:
            print('mAP: %.4f' % summary['mean_ap'])
            names = {'trans_err': 'mATE', 'scale_err': 'mASE', 'orient_err': 'mAOE', 'vel_err': 'mAVE', 'attr_err': 'mAAE'}
            for k, v in summary['tp_errors'].items():
                print('%s: %.4f' % (names[k], v))
            print('mRec: %.4f' % summary['mean_recall'])
            print('NDS: %.4f' % summary['nd_score'])
            print('Eval time: %.1fs' % summary['eval_time'])
            print()
            print('Per-class results:')
            print('%-20s\t%-6s\t%-6s\t%-6s\t%-6s\t%-6s\t%-6s\t%-6s' % ('Object Class', 'AP', 'ATE', 'ASE', 'AOE', 'AVE', 'AAE', 'avgRec'))
            for i, name in enumerate(summary['mean_dist_aps']):
                t = summary['label_tp_errors'][name]
                print('%-20s\t%-6.3f\t%-6.3f\t%-6.3f\t%-6.3f\t%-6.3f\t%-6.3f\t%-6.3f' % (name, summary['mean_dist_aps'][name], t['trans_err'],
                                                                                         t['scale_err'], t['orient_err'], t['vel_err'],
                                                                                         t['attr_err'], recall_list[i]))
        return summary
