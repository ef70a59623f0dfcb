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


def test_bench_self_launches_its_ranks_and_propagates_their_failure(has_gpu):
    """`python bench.py --gpus 2` is ONE command: the parent starts the launcher + ranks as children; without a GPU the ranks refuse
    (no CPU fallback) and the parent's exit code says so."""
    if has_gpu:
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "1024"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_hop_probe_on_the_host_clock():
    import torch
    from bachelor_thesis_n_body_amd.dist import HopProbe

    class Req:
        def wait(self):
            pass
    p = HopProbe(torch, torch.device("cpu"))
    p.enabled = True
    ran = []
    for _ in range(3):
        p.posting()
        p.exchange([Req(), Req()], lambda: ran.append(1), host_blocking=True)
    s = p.summary()
    assert len(ran) == 3 and s["hops_timed"] == 3 and s["comm_ms"] >= 0 and s["comm_ms_max"] >= s["comm_ms"]
