#!/usr/bin/env python3
"""Extracts CONCEPT's actual time-step schedule from the run logs the reference ships:

    TheFeedingPipeline/Datafolder4/{Test_Data_concept,Til_rapport ...}/id*/params_id*.log

Each base step of the main loop prints (e.g. params_id0200_....log:49-55, 89-96)
    Time step 17 / Scale factor: 0.02512 / Cosmic time: 0.06750 Gyr / Step size: 0.006071 Gyr / Rung population: matter: 32760, 8
Output (committed beside this script, the script itself needs /root/reference and is not run by the tests):
    concept_step_schedule.txt  — columns: step, a (scale factor at the START of the step), cosmic time t [Gyr], step size dt [Gyr],
                                 then over all logs found: min / max of a at that step index, the mean fraction of particles above rung 0, the number of logs, and the mean populations of rungs 0..4
    concept_step_schedule_runs.txt — per log: id, A_s, number of base steps, a at the last step, largest rung ever populated
Usage: python make_step_schedule.py [/root/reference]"""
import re
import sys
from pathlib import Path

import numpy as np

ANSI = re.compile(r"\x1b\[[0-9;]*[A-Za-z]|\x1b\][^\\]*\\|\x1b\(B")


def parse_log(path):
    """-> list of (step, a, t, dt, [rung populations])"""
    txt = ANSI.sub("", Path(path).read_text(encoding="utf8", errors="replace"))
    steps = []
    for m in re.finditer(r"Time step (\d+)\s*\nScale factor:\s*([0-9.eE+-]+)\s*\nCosmic time:\s*([0-9.eE+-]+)\s*Gyr\s*\nStep size:\s*([0-9.eE+-]+)\s*Gyr[^\n]*\n"
                         r"Rung population:\s*\n\s*matter:\s*([0-9, ]+)", txt):
        rungs = [int(v) for v in m.group(5).replace(" ", "").split(",") if v]
        steps.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), rungs))
    return steps


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    logs = sorted((ref / "TheFeedingPipeline").rglob("params_id*.log"))
    runs = []
    for lg in logs:
        st = parse_log(lg)
        if not st:
            continue
        m = re.search(r"id(\d+)_A_s([0-9.]+e[+-]?\d+)", lg.name)
        runs.append((int(m.group(1)), float(m.group(2)), st, lg.parent.parent.name))
    if not runs:
        raise SystemExit("no logs found")
    nmax = max(len(r[2]) for r in runs)
    here = Path(__file__).resolve().parent
    # the schedule of the first TEST job (id0200, the log SURVEY.md cites), with the spread over all runs
    base = next(r for r in runs if r[0] == 200)[2]
    rows = []
    for i, (step, a, t, dt, rungs) in enumerate(base):
        a_all = np.array([r[2][i][1] for r in runs if len(r[2]) > i])
        hi = np.array([1.0 - r[2][i][4][0] / sum(r[2][i][4]) for r in runs if len(r[2]) > i])
        pops = np.zeros(5)
        for r in runs:
            if len(r[2]) > i:
                for l, v in enumerate(r[2][i][4][:5]):
                    pops[l] += v
        rows.append([step, a, t, dt, a_all.min(), a_all.max(), hi.mean(), len(a_all)] + list(pops / len(a_all)))
    np.savetxt(here / "concept_step_schedule.txt", np.array(rows), fmt=["%d", "%.5f", "%.5f", "%.6f", "%.5f", "%.5f", "%.5f", "%d"] + ["%.2f"] * 5,
               header="CONCEPT base time steps of job id0200 (TheFeedingPipeline/Datafolder4/Test_Data_concept/id0200_*/params_id0200_*.log), a = 0.02 -> 1, 32^3 p3m\n"
                      "step  a_start  t[Gyr]  dt[Gyr]  min_a_over_runs  max_a_over_runs  mean_fraction_above_rung0  runs_with_this_step  mean populations of rungs 0..4 (of 32768)\n"
                      f"{len(runs)} logs parsed; longest run {nmax} steps")
    with open(here / "concept_step_schedule_runs.txt", "w") as f:
        f.write("# folder id A_s base_steps a_last_step max_rung\n")
        for rid, As, st, folder in runs:
            f.write(f"{folder} {rid} {As:.10e} {len(st)} {st[-1][1]:.5f} {max(len(s[4]) for s in st) - 1}\n")
    print(f"{len(runs)} logs; steps per run min/max {min(len(r[2]) for r in runs)}/{nmax}")


if __name__ == "__main__":
    main()
