"""bench.py contract pieces that can be checked without a GPU: it refuses to run without a device (no CPU fallback), and its
cpu_baseline leg times the C oracle on a bounded sample and reports the fields the driver expects."""
import json
import os
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
    assert "back_comm_ms" not in s                     # one-sided schedule: no return leg
    # the return leg of the pair-symmetric schedule: post -> arrival, the waits for returned partials, the closing adds
    for _ in range(2):
        reqs = [Req()]
        p.returning(p._mark(), reqs, host_blocking=True)
        p.wait_back(reqs)
        p.closing(lambda: ran.append(2))
    s = p.summary()
    assert s["returns_timed"] == 2 and s["evaluations_timed"] == 2 and s["back_comm_ms"] >= 0 and s["back_exposed_ms"] >= 0 and s["add_ms"] >= 0
    assert ran.count(2) == 2


STALL_CHILD = """
import os, sys, time, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode, stall_step = sys.argv[1], int(sys.argv[2])
hb = bench.Heartbeat(rank)
done = {}
def on_stall(label):
    if done:
        if rank == 0:
            print(json.dumps({"value": done["first"], "stalled": {"at": label}}), flush=True)
        return 0
    return 3
wd = bench.StallWatchdog(hb, bench.stall_seconds(), on_stall) if mode == "inprocess" else None
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.ones(4)
for region in ("first", "second"):
    for k in range(5):
        if region == "second" and k == stall_step and rank == 1:
            time.sleep(1e6)                      # injected stall: this rank stops taking part, the other waits in the collective
        dist.all_reduce(t)
        hb("%%s: step %%d" %% (region, k + 1))
    done[region] = float(t.sum())
if wd: wd.stop()
if rank == 0:
    print(json.dumps({"value": done["second"]}), flush=True)
dist.destroy_process_group()
"""


def _launch_stall_child(tmp_path, mode, stall_step, stall_s, margin_s, capfd):
    import argparse
    import bench
    script = tmp_path / "stall_child.py"
    script.write_text(STALL_CHILD % str(ROOT))
    old = {k: os.environ.get(k) for k in ("NB_BENCH_STALL_S", "NB_BENCH_PARENT_MARGIN_S")}
    os.environ.update(NB_BENCH_STALL_S=str(stall_s), NB_BENCH_PARENT_MARGIN_S=str(margin_s))
    try:
        rc = bench.self_launch(argparse.Namespace(gpus=2), argv=[mode, str(stall_step)], child=str(script))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    out = capfd.readouterr()
    return rc, out.out, out.err


def test_stall_watchdog_reports_the_completed_region_when_a_collective_deadlocks(tmp_path, capfd):
    """Two gloo ranks, all alive, rank 1 stops taking part in the second region: every rank's own StallWatchdog fires, rank 0 prints the
    line for the region that completed (naming where the other stopped) and the command returns 0 — bench.py's N > 1 behaviour."""
    rc, out, err = _launch_stall_child(tmp_path, "inprocess", 2, 3, 60, capfd)
    lines = [json.loads(l) for l in out.splitlines() if l.strip().startswith("{")]
    assert rc == 0 and len(lines) == 1, (rc, out, err[-2000:])
    assert lines[0]["value"] == 4.0 * 2 ** 5 and lines[0]["stalled"]["at"] == "second: step 2"


def test_parent_supervisor_kills_all_ranks_when_none_advances(tmp_path, capfd):
    """The same deadlock without the in-process watchdog: the parent sees no heartbeat file move, kills both ranks by PID, names the
    last mark of each and returns 4 instead of waiting for the caller's time limit."""
    rc, out, err = _launch_stall_child(tmp_path, "parent", 3, 2, 1, capfd)
    assert rc == 4 and not [l for l in out.splitlines() if l.strip().startswith("{")], (rc, out, err[-2000:])
    assert "no rank advanced" in err and "rank 0: second: step 3" in err and "rank 1: second: step 3" in err


def test_a_healthy_multi_rank_run_is_not_disturbed_by_the_watchdogs(tmp_path, capfd):
    rc, out, err = _launch_stall_child(tmp_path, "inprocess", 99, 3, 1, capfd)
    lines = [json.loads(l) for l in out.splitlines() if l.strip().startswith("{")]
    assert rc == 0 and len(lines) == 1 and "stalled" not in lines[0] and lines[0]["value"] == 4.0 * 2 ** 10


def test_host_identity_and_numpy_leg():
    import bench
    import bachelor_thesis_n_body_amd as nb
    h = bench.host_identity(3)
    assert h["cpu_count"] >= 1 and h["cpus_available_to_this_process"] >= 1 and h["omp_threads"] == 3 and "model" in h
    s = nb.plummer_sphere(2048, 0, np.float32)
    r = bench.cpu_baseline_numpy(s, 0.01, n_src=1024, budget_s=0.2)
    assert r["kind"] == "port" and r["cores"] == 1 and r["value"] > 1e5 and "NumPy oracle" in r["sample"]
    r = bench.cpu_baseline(s, 0.01, budget_s=0.2, numpy_budget_s=0.2)
    assert r["host"]["omp_threads"] == r["cores"] and r["numpy"]["value"] > 0
