#!/usr/bin/env python3
"""Entry point of the detection evaluation (reference src/nuscenes/eval_custom.py:1158-1214): same positional argument
and flags; the tables are read by cm3d_amd.nusc_io.NuscTables instead of nuscenes-devkit, so `--eval_set` is either
`all` (every scene of the tables, the default), a comma-separated list of scene names, or a JSON file holding such a list.
Plot / render flags are accepted for command-line compatibility and ignored (no matplotlib dependency here)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from cm3d_amd import eval_detection as ev, nusc_io  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser(description='Evaluate nuScenes detection results.',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('result_path', type=str, help='The submission as a JSON file.')
    parser.add_argument('--output_dir', type=str, default='~/nuscenes-metrics')
    parser.add_argument('--eval_set', type=str, default='all')
    parser.add_argument('--dataroot', type=str, default='/data/sets/nuscenes')
    parser.add_argument('--version', type=str, default='v1.0-trainval')
    parser.add_argument('--config_path', type=str, default='')
    parser.add_argument('--plot_examples', type=int, default=0)
    parser.add_argument('--render_curves', type=int, default=0)
    parser.add_argument('--verbose', type=int, default=1)
    parser.add_argument('--drivable_filtering', type=int, default=0)
    parser.add_argument('--object_only', type=int, default=0)
    args = parser.parse_args(argv)
    if args.config_path == '':
        cfg = ev.config_factory('detection_cvpr_2019')
    else:
        with open(args.config_path) as f:
            cfg = ev.DetectionConfig.deserialize(json.load(f))
    if args.eval_set == 'all':
        scenes = None
    elif os.path.exists(args.eval_set):
        with open(args.eval_set) as f:
            scenes = set(json.load(f))
    else:
        scenes = set(args.eval_set.split(','))
    tables = nusc_io.NuscTables(args.version, os.path.expanduser(args.dataroot))
    de = ev.DetectionEval(tables, cfg, os.path.expanduser(args.result_path), scenes, os.path.expanduser(args.output_dir),
                          bool(args.drivable_filtering), bool(args.object_only), bool(args.verbose))
    return de.main()


if __name__ == "__main__":
    main()
