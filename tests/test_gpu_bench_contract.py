"""bench.py end to end on the GPU at a reduced size: ONE JSON line with every key of the driver's contract plus the roofline
and cpu_baseline objects."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_prints_one_contract_json_line():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--n", "65536"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] == pytest.approx(65536 ** 2 * 3 / (j["ms_per_step"] * 3e-3), rel=1e-6)
    r = j["roofline"]
    assert r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.2 < r["frac"] < 1.0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["value"] < j["value"]
