#!/usr/bin/env python3
"""Host side of the nuScenes entry point (file reads, RLE strings, packing: pipeline_nuscenes.prepare_scene_batch) on a
synthetic C1-shaped dataset written to a temporary directory: frames/s in this process and with N spawned reader
processes.  CPU only."""
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3d_amd import nusc_io, pipeline_nuscenes as pn, synthetic as syn  # noqa: E402


def main():
    cfg = syn.config("c1")
    d = tempfile.mkdtemp(prefix="cm3d_readers_")
    try:
        n_scenes, fps = 8, 6
        dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(d, cfg, n_scenes=n_scenes, frames_per_scene=fps)
        tasks = [("v1.0-synth", dataroot, mask_dir, [n], 3, cfg.ratio, False, None) for n in names]
        t0 = time.time()
        for t in tasks:
            pn.prepare_scene_batch(t)
        print(f"in this process: {n_scenes * fps / (time.time() - t0):.0f} frames/s")
        for w in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
            with mp.get_context("spawn").Pool(w) as pool:
                def consume(it):
                    keep = []
                    for _, batches, _ in it:                              # what lift_scenes does with a prepared batch
                        for hb in batches:
                            pn._attach_raw(hb, keep)
                            hb.raw.sum()                                 # touch the mapped sweeps
                            pn._release(keep)
                shm_tasks = [t + (True,) for t in tasks]
                for _ in range(3):                                      # start-up (imports, tables) outside the timing
                    consume(pool.imap(pn.prepare_scene_batch, shm_tasks))
                t0 = time.time()
                consume(pool.imap(pn.prepare_scene_batch, shm_tasks))
                dt = time.time() - t0
            print(f"{w} reader processes: {n_scenes * fps / dt:.0f} frames/s")
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
