#!/usr/bin/env python3
"""Where a step's time goes at launch-sensitive sizes: `rocprofv3 --kernel-trace` of bench.py at --n N, then per step (one pair-symmetric force
launch to the next) the kernels' durations and the idle gaps between them.
Usage (GPU box, repo root): python3 tools/step_timeline.py [N] [steps] [extra bench.py args]   -> prints a summary; trace under gpurun_out/step_timeline"""
import csv, glob, os, subprocess, sys
n = sys.argv[1] if len(sys.argv) > 1 else "65536"
steps = sys.argv[2] if len(sys.argv) > 2 else "30"
extra = sys.argv[3:]
out = "gpurun_out/step_timeline"
subprocess.run(["rm", "-rf", out]); os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, "bench.py", "--n", n, "--steps", steps, "--warmup", "3",
       "--no-cpu-baseline", "--no-clock-probe", "--traffic", "none"] + extra
r = subprocess.run(cmd, env=env, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stdout[-2000:] + r.stderr[-2000:])
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
rows.sort()
def short(name):
    for key in ("accel_sym_boundary", "accel_sym_reduce", "accel_sym_tail", "accel_sym_f", "kick_drift", "kick_kernel", "drift_kernel", "accel_f"):
        if key in name:
            return key
    return name[:40]
force = [i for i, r in enumerate(rows) if short(r[2]) == "accel_sym_f"]
if len(force) < 8:
    sys.exit("fewer than 8 pair-symmetric launches in the trace")
force = force[-(int(steps) - 1):]          # the timed steps
period, busy, gaps = [], {}, []
for a, b in zip(force[:-1], force[1:]):
    period.append(rows[b][0] - rows[a][0])
    g = 0
    for i in range(a, b):
        busy.setdefault(short(rows[i][2]), []).append(rows[i][1] - rows[i][0])
        g += max(0, rows[i + 1][0] - rows[i][1])
    gaps.append(g)
mean = lambda v: sum(v) / len(v)
print(f"n={n}: {len(period)} steps, force launch to force launch {mean(period) / 1e3:.1f} us (min {min(period) / 1e3:.1f})")
for k, v in sorted(busy.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:22s} {len(v) / len(period):.1f} per step, mean {mean(v) / 1e3:7.1f} us, per step {sum(v) / len(period) / 1e3:7.1f} us")
print(f"  idle between kernels    per step {mean(gaps) / 1e3:7.1f} us")
