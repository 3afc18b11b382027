"""CPU: `python bench.py --gpus N` brings up its N ranks by itself (no torchrun wrapper), and refuses a rank count that
does not match --gpus.  The rehearsal mode runs the launch, the rendezvous and the single record exchange without kernels."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CM3D_DIST_BACKEND="gloo", **extra)
    return env


def test_bench_launches_its_own_ranks():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rehearse-launch"], cwd=ROOT, env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_group"] == 2 and d["gather_ok"] is True and d["backend"] == "gloo"


def test_bench_launches_eight_ranks_for_the_drivers_scaling_run():
    """The command line the driver uses at N = 8 (`python bench.py --gpus 8 ...`), rehearsed on the CPU: eight ranks come up over
    gloo, rendezvous on 127.0.0.1, exchange their records once, rank 0 prints the line."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--rehearse-launch"], cwd=ROOT, env=_env(OMP_NUM_THREADS="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_in_group"] == 8 and d["gather_ok"] is True


def test_bench_refuses_wrong_world_size():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--rehearse-launch"], cwd=ROOT,
                       env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()


def test_byte_accounting_counts_only_bytes_that_move():
    sys.path.insert(0, ROOT)
    import bench
    # a 1600x900 mask whose eroded rectangle is 100x50 pixels: 5 words x 50 rows exist, not 50 x 900
    bbox = np.array([[64, 10, 163, 59], [0x7FFFFFFF, 0x7FFFFFFF, -1, -1]], np.int32)
    assert bench.packed_rect_bytes(bbox, 50) == 4 * (163 // 32 - 64 // 32 + 1) * 50

    class HB:
        n_frames, n_masks, width, height, raw_stride, n_raw_rows, bytes_per_row = 2, 2, 1600, 900, 5, 70000, 20
        rle_counts = np.zeros(1000, np.uint32)

    class HBQ(HB):              # the quad layout: 12 bytes of a row cross HBM
        raw_stride, bytes_per_row = 3, 12
    assert bench.compulsory_bytes(HBQ, 1, 5000, "rle", True, False, 1000)["k_project_hits"] == 70000 * (12 + 4)
    by = bench.compulsory_bytes(HB, 1, 5000, "rle", True, False, 1000)
    assert by["k_project_hits"] == 70000 * (20 + 4)                       # raw rows in, hit words out, no cloud, no phantom masks
    assert bench.compulsory_bytes(HB, 1, 5000, "rle", True, True, 1000)["k_project_hits"] == 70000 * (20 + 4 + 16)
    assert bench.compulsory_bytes(HB, 1, 5000, "rle", False, True, 1000)["k_project_hits"] == 70000 * (16 + 4)
    assert by["k_rle_erode_pack_wave"] == 4000 + 1000
    full_masks = 2 * 50 * 4 * 900
    assert by["pass_total"] < full_masks + by["k_project_hits"] + 200000     # nothing near a full read of the packed masks
    assert bench.visible_cores() >= 1
