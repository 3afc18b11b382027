#!/usr/bin/env python3
"""Runs ON THE GPU BOX: what each stage costs WITH THREE BATCHES IN FLIGHT.  Three engines with their own batch and stream run
passes round-robin (as bench.py does); one stage at a time is left out of every pass.  The batches stay resident, so a stage that
is left out leaves the results of the earlier passes in place and everything behind it does the same work as before: the drop of
the time per pass is what the stage costs in flight (its alone-time is tools/stage_time.py's).
usage: tools/stage_ablate.py [config frames [passes]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import lifting, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 300
DEPTH = 3
cfg = syn.config(name)
engs, streams = [], []
for d in range(DEPTH):
    frames = [syn.make_frame(cfg, 1000 * d + i) for i in range(F)]
    lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
    hb = lifting.pack_frames(frames, lanes, [0] * F)
    del frames
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        e = lifting.LiftEngine()
        e.upload(hb)
        for _ in range(3):
            e.run(masks="rle")
    engs.append(e); streams.append(s)
    print(f"batch {d} resident", flush=True)
torch.cuda.synchronize()
for e in engs:
    e.check_status()


def one_pass(e, skip):
    st = torch.cuda.current_stream().cuda_stream
    if "begin+project" not in skip:
        e.stage_begin(st)
    if "masks" not in skip:
        e.stage_masks(st, "rle")
    if "begin+project" not in skip:
        e.stage_sweep_project(st, None)
    if "compact" not in skip:
        e.stage_compact(st)
    if "medoid" not in skip:
        e.stage_medoid(st)
    if "lanes" not in skip:
        e.wait_lane_grid()
        e.stage_lanes(st)
    if "boxes" not in skip:
        e.stage_boxes(st)


def timed(skip, n=N):
    for k in range(30):
        with torch.cuda.stream(streams[k % DEPTH]):
            one_pass(engs[k % DEPTH], skip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        with torch.cuda.stream(streams[k % DEPTH]):
            one_pass(engs[k % DEPTH], skip)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


base = timed(())
print(f"{name} x{F}, {DEPTH} batches in flight: full pass {base:.1f} us")
# parts of the mask kernel (CM3D_RLE_DIAG, read per call): without its stores the packed masks of the earlier passes stay valid
# (only bit 1 keeps what follows valid: the other switches leave empty bounding boxes behind)
for bits, what in ((1, "the mask kernel's stores"),):
    os.environ["CM3D_RLE_DIAG"] = str(bits)
    t = timed(())
    print(f"  without {what:44s} {t:7.1f} us   ({base - t:+6.1f})", flush=True)
os.environ["CM3D_RLE_DIAG"] = "0"
for skip in (("masks",), ("begin+project",), ("compact",), ("medoid",), ("lanes",), ("boxes",), ("lanes", "boxes"), ("masks", "compact", "medoid", "lanes", "boxes"), ()):
    t = timed(skip)
    print(f"  without {' + '.join(skip) if skip else 'nothing (again)':44s} {t:7.1f} us   ({base - t:+6.1f})", flush=True)
