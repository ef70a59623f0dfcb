"""bench.py contract pieces that can be checked without a GPU: it refuses to run without a device (no CPU fallback), and its
cpu_baseline leg times the C oracle on a bounded sample and reports the fields the driver expects."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_bench_refuses_to_run_without_gpu(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_cpu_baseline_leg():
    import bench
    import bachelor_thesis_n_body_amd as nb
    s = nb.plummer_sphere(4096, 0, np.float32)
    r = bench.cpu_baseline(s, 0.01, budget_s=0.3)
    assert r["kind"] == "port" and r["unit"] == "interactions/s" and 1 <= r["cores"] <= 16
    assert r["value"] > 1e6 and "sources" in r["sample"]


def test_measured_traffic_reads_the_committed_profile():
    import bench
    t = bench.measured_traffic()
    assert t is None or 1e6 < t < 1e11
