"""The shader clock under the headline kernel, step by step (round 4; VERDICT r3 "reconcile the clock story").

bench.py reports ONE sustained clock per timed region; round 3's records disagreed (2 012 MHz over a 5-step window, 2 146 MHz over a
20-step window of the same binary, while ms/step barely moved). This tool opens one probe window PER STEP — the first step reported
separately — for a 5-step and a 20-step run in one process on one box, each XCD's clock separately (the probe runs one wave per XCD),
Prints JSON lines. (A window over the whole run cannot be held open beside the per-step ones: releasing a probe frees its device buffer, which synchronises the device.)   usage: python tools/clock_trace.py [n] [f32|f64]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
eps, dt = 0.01, 1.0 / 256
torch.cuda.set_device(0)
eng = nb.ShardedEngine(n, dtype, device=torch.device("cuda", 0))
eng.set_state(nb.plummer_sphere(n, 0, dtype))
eng.accel(eps)
torch.cuda.synchronize()
stream = lambda: torch.cuda.current_stream().cuda_stream
for steps in (5, 20):
    time.sleep(2.0)                       # let the board idle between the runs, as between two bench invocations
    rows, t_run = [], time.perf_counter()
    for k in range(steps):
        p = nb.ClockProbe(0, 5000.0)
        t0 = time.perf_counter()
        eng.step(dt, 1, eps)
        p.stop(stream())
        torch.cuda.current_stream().synchronize()      # NOT a device-wide synchronisation: that would wait for the whole-run probe's window
        ms = (time.perf_counter() - t0) * 1e3
        mhz, win, xcd = p.end_xcd()
        rows.append({"step": k, "ms": ms, "mhz": mhz, "window_ms": win, "xcd_mhz": [round(v, 1) for v in xcd],
                     "cycles_per_64": ms * 1e-3 * mhz * 1e6 * 1024 / (float(n) * n / 64)})
    rest = rows[1:]
    print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "steps": steps, "first_step": rows[0],
                      "other_steps_mean_ms": float(np.mean([r["ms"] for r in rest])), "other_steps_mean_mhz": float(np.mean([r["mhz"] for r in rest])),
                      "other_steps_min_max_mhz": [float(min(r["mhz"] for r in rest)), float(max(r["mhz"] for r in rest))],
                      "other_steps_mean_cycles_per_64": float(np.mean([r["cycles_per_64"] for r in rest])),
                      "xcd_mhz_mean_over_other_steps": [round(float(np.mean([r["xcd_mhz"][i] for r in rest])), 1) for i in range(len(rest[0]["xcd_mhz"]))],
                      "time_weighted_mean_mhz_all_steps": float(np.sum([r["mhz"] * r["ms"] for r in rows]) / np.sum([r["ms"] for r in rows])),
                      "wall_ms": (time.perf_counter() - t_run) * 1e3,
                      "per_step_mhz": [round(r["mhz"], 1) for r in rows], "per_step_ms": [round(r["ms"], 3) for r in rows]}), flush=True)
