"""GPU: bench.py prints ONE JSON line with the fields the driver and the tier contract name (small workload)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, "bench.py", *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_lifting_bench_line():
    d = _run("--config", "tiny", "--frames", "8", "--steps", "6", "--warmup", "2", "--cpu-sample", "2", "--cpu-workers", "2",
             "--lane-points", "2000", "--e2e-frames", "64")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    assert "workload" in d["config"] and d["config"]["batches_in_flight"] == 4
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "traffic" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["avg_launch_ms"] > 0
    # bytes = what must move: every raw row (12 B in the quad layout the bench packs) + a hit word (4 B) for the rows of 256-row blocks
    # that hold an in-mask point -- the only hit words the launch writes; nothing the kernel skips is counted; no fraction anywhere above 1
    rows, nbytes = rf["rows_per_launch"], rf["bytes_per_launch"]
    assert (nbytes - 12 * rows) % 4 == 0 and 0 < (nbytes - 12 * rows) // 4 <= rows
    assert abs(rf["bytes_per_row"] - nbytes / rows) < 0.01 and 12 < rf["bytes_per_row"] <= 16
    # the K timed steps against five more regions of the same K steps and one long region: the line's value must be
    # one of that family (30 % slack on either side: six passes over eight tiny frames are a few hundred microseconds)
    sp = d["value_spread"]
    assert sp["repeats"] == 5 and sp["min"] <= sp["median"] <= sp["max"] and 0.7 * sp["min"] <= d["value"] <= 1.3 * sp["max"]
    assert d["value_long"] > 0 and d["value_long_steps"] >= 6
    # files -> labels through the entry point, timed inside the default line
    e2e = d["end_to_end"]
    assert "error" not in e2e and e2e["frames"] == 64 and e2e["frames_per_s"] > 0 and e2e["boxes"] > 0 and e2e["timer"]["total"] > 0
    md = d["kernels"]["medoid"]
    assert md["bound"] == "valu" and 0 < md["frac"] < 1 and 3.5 <= md["algorithmic_slots_per_pair"] <= 11.0
    assert d["config"]["cloud_materialised"] is False and d["ranks_in_group"] == 1

    def fracs(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k.startswith("frac") and isinstance(v, (int, float)):
                    yield v
                yield from fracs(v)
    assert all(0 <= v <= 1.0 for v in fracs(d)), list(fracs(d))
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 20 * cb["value"]
    one = _run("--config", "tiny", "--frames", "8", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-secondary", "--in-flight", "1",
               "--lane-points", "2000")
    assert one["config"]["batches_in_flight"] == 1 and one["cpu_baseline"] is None


def test_two_ranks_from_a_bare_command_line():
    """`python bench.py --gpus 2` with no torchrun wrapper: two ranks (sharing this box's one GPU, gloo for the exchange)."""
    env = dict(os.environ, CM3D_SINGLE_DEVICE="1", CM3D_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "tiny", "--frames", "8", "--steps", "4", "--warmup", "1",
                        "--cpu-sample", "0", "--no-secondary", "--lane-points", "2000"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_group"] == 2 and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_fusion_bench_line():
    d = _run("--fusion", "300", "--steps", "5", "--warmup", "1")
    assert d["unit"] == "samples/s" and d["equals_oracle"] is True and d["matches"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
