"""bench.py end to end on the GPU at a reduced size: ONE JSON line with every key of the driver's contract plus the roofline
and cpu_baseline objects."""
import json
import os
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
    # both CPU legs and the host they ran on
    assert c["host"]["cpu_count"] >= c["cores"] == c["host"]["omp_threads"] and isinstance(c["host"]["model"], str)
    assert c["numpy"]["cores"] == 1 and 0 < c["numpy"]["value"] < c["value"] and "16384 sources" in c["numpy"]["sample"]
    assert r["variant"] == "uniform_mass" and j["world_size"] == 1 and j["ring"] is None
    # HBM traffic measured live (two rocprofv3 --pmc child passes of the same command), not read from a committed file
    assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
    assert r["traffic"] > r["algorithmic_bytes_per_launch"] == 28 * 65536
    # the shader clock held during the timed region, and the box-independent cycles per 64 interactions and SIMD (issue floor: 30)
    assert 1200.0 < r["sustained_clock_mhz"] < 2600.0 and r["clock_window_ms"] > 0
    assert r["frac_at_sustained_clock"] == pytest.approx(r["frac"] * 2400.0 / r["sustained_clock_mhz"])
    # N = 65536 takes the pair-symmetric kernel (every unordered pair once): issue floor ~20 cycles per 64 delivered interactions (one-sided: 30)
    assert r["pair_symmetry"] is True and r["kernel"] == "accel_sym_f32_kernel" and r["executed_flop_per_interaction"] == 12.0
    assert r["frac_executed_flops"] == pytest.approx(r["frac"] * 12.0 / 20.0)
    assert 17.0 < r["cycles_per_64_interactions_per_simd"] < 80.0


def test_bench_general_mass_variant_is_labelled():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--n", "32768", "--general-mass", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert j["roofline"]["variant"] == "general" and "general" in j["config"]["masses"]


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_bench_two_ranks_self_launched_on_one_gpu(dtype):
    """`python bench.py --gpus 2` as ONE command: the parent spawns the launcher and the two ranks, which share the test box's one GPU
    (gloo transport, --same-device) — the bench's barrier, MAX all-reduce, rank-0 print and per-hop ring timing end to end."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--n", "65536", "--dtype", dtype,
                          "--backend", "gloo", "--same-device"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stderr[-3000:], out.stdout[-1000:])
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["backend"] == "gloo" and j["same_device"] is True and j["scaling"] == "strong"
    assert j["dtype"] == dtype and j["value"] == pytest.approx(65536 ** 2 * 3 / (j["ms_per_step"] * 3e-3), rel=1e-6)
    ring = j["ring"]
    assert ring["hops_timed"] == 3 and ring["comm_ms"] > 0 and ring["kernel_ms"] > 0 and ring["exposed_ms"] >= 0
    r = j["roofline"]
    assert r["launches_timed"] == 6 and r["interactions_per_launch"] == 32768.0 ** 2 and 0.05 < r["frac"] < 1.0
    assert "cpu_baseline" not in j          # rank 0 at N = 1 only
    assert 1200.0 < r["sustained_clock_mhz_min_over_ranks"] <= r["sustained_clock_mhz"] * 1.2
    # BOTH schedules were timed in the one invocation (3 steps each); the contract keys are the faster one's
    one, pair = r["one_sided"], r["pair_symmetric"]
    for blk in (one, pair):
        assert blk["launches_timed"] == 6 and blk["ms_per_step"] > 0 and blk["kernel_ms_per_step"] > 0 and blk["ring"]["hops_timed"] == 3
        assert blk["value"] == pytest.approx(65536 ** 2 * 3 / (blk["ms_per_step"] * 3e-3), rel=1e-6)
    assert j["ms_per_step"] == pytest.approx(min(one["ms_per_step"], pair["ms_per_step"]))
    assert ("pair-symmetric" in r["schedule"]) == (pair["ms_per_step"] <= one["ms_per_step"]) and "stalled" not in r
    # the return leg of the pair-symmetric schedule is visible: one partial per step (the split half-way hop of the 2-ring), waits, closing adds
    pr = pair["ring"]
    assert pr["returns_timed"] == 3 and pr["evaluations_timed"] == 3 and pr["back_comm_ms"] > 0 and pr["back_exposed_ms"] >= 0 and pr["add_ms"] > 0
    assert "back_comm_ms" not in one["ring"]


def test_bench_reports_the_completed_schedule_when_the_other_stalls():
    """All ranks alive, rank 1 stops taking part in the SECOND schedule's timed region (injected): each rank's watchdog fires after
    NB_BENCH_STALL_S, rank 0 prints the line of the schedule that completed — labelled — and the command returns 0."""
    env = dict(os.environ, NB_BENCH_STALL_S="15", NB_BENCH_PARENT_MARGIN_S="60")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--n", "32768",
                          "--backend", "gloo", "--same-device", "--inject-stall", "1:1:1"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stderr[-3000:], out.stdout[-1000:])
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    r = j["roofline"]
    assert r["one_sided"] is not None and r["pair_symmetric"] is None and "one-sided" in r["schedule"]
    assert r["stalled"]["schedules_not_completed"] == ["pair-symmetric"] and "pair-symmetric" in r["stalled"]["at"]
    assert j["ms_per_step"] == pytest.approx(r["one_sided"]["ms_per_step"]) and j["value"] > 0


def test_bench_single_schedule_switch():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "32768",
                          "--backend", "gloo", "--same-device", "--schedule", "pair-symmetric"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stderr[-3000:], out.stdout[-1000:])
    r = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])["roofline"]
    assert r["one_sided"] is None and r["pair_symmetric"]["launches_timed"] == 4 and "pair-symmetric" in r["schedule"]


def test_bench_propagates_a_failing_rank():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "4096"],
                         capture_output=True, text=True, timeout=600)      # RCCL on a one-GPU box: rank 1 has no device
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_single_process_multi_device_context():
    """`--single-process`: the N-GPU workload through nb_create_multi (one process, peer pulls), rehearsed with both shards on GPU 0."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--single-process", "--same-device", "--steps", "3", "--warmup", "1",
                          "--n", "65536"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert j["n_gpus"] == 2 and "nb_create_multi" in j["config"]["parallelism"] and j["value"] == pytest.approx(65536 ** 2 * 3 / (j["ms_per_step"] * 3e-3), rel=1e-6)
    assert 0.05 < j["roofline"]["frac"] * 2 < 1.0          # two shards share one GPU here


def test_clock_probe_measures_a_plausible_shader_clock():
    """nb_clock_probe_begin / _end: one sleeping wave, cycles over the constant-rate counter; beside a running all-pairs launch."""
    import numpy as np
    import bachelor_thesis_n_body_amd as nb
    n = 1 << 17
    with nb.Engine(n, np.float32) as e:
        e.set_state(nb.plummer_sphere(n, 0, np.float32))
        e.accel(0.01, fetch=False)
        probe = nb.ClockProbe(0, 20.0)                  # a fixed window of 20 ms
        for _ in range(6):
            e.accel(0.01, fetch=False)
        mhz, window = probe.end()
        assert 1200.0 < mhz < 2600.0 and 20.0 <= window < 40.0
        probe = nb.ClockProbe(0, 3000.0)                # a window ended from a stream: as long as the work before the stop, not 3 s
        for _ in range(6):
            e.accel(0.01, fetch=False)
        probe.stop(0)
        mhz, window = probe.end()
        assert 1200.0 < mhz < 2600.0 and 1.0 < window < 1000.0
    with pytest.raises(RuntimeError):
        probe.end()
    with pytest.raises(RuntimeError):
        nb.ClockProbe(99, 1.0)
    with pytest.raises(RuntimeError):
        nb.ClockProbe(0, 0.0)
