"""One process of bench.py's all-cores CPU baseline (test/measurement infrastructure, like everything under
oracle/): builds its own synthetic frames -- the generator is deterministic per frame index -- runs the C oracle
over them and prints {"frames": n, "seconds": oracle time} as one JSON line.

usage: python -m oracle.cpu_worker <config> <first_frame> <count> <lane_points> [KEY=VALUE ...]"""
import json
import sys
import time


def main(argv):
    from cm3d_amd import lifting, synthetic as syn
    from oracle import oracle as orc
    from tests.helpers import oracle_batch
    name, first, count, lane_points = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    over = {}
    for kv in argv[4:]:
        k, v = kv.split("=", 1)
        cur = getattr(syn.SyntheticConfig(), k)
        over[k] = v if isinstance(cur, str) else type(cur)(float(v))
    cfg = syn.config(name, **over)
    orc.lib()
    frames = [syn.make_frame(cfg, first + i) for i in range(count)]
    lanes = [syn.make_lane_table([600.0, 1600.0], lane_points, seed=7, extent=260.0)]
    fl = [0] * count
    hb = lifting.pack_frames(frames, lanes, fl)
    t0 = time.perf_counter()
    oracle_batch(orc, frames, lanes, fl, hb)
    print(json.dumps({"frames": count, "seconds": time.perf_counter() - t0}))


if __name__ == "__main__":
    main(sys.argv[1:])
