"""Scan of the rung limiter's constant eta (and of the NB_RUNG_SR_ONLY variant: limiter from the pair force alone instead of the total acceleration) against the rung
populations the reference's logs record (tests/golden/concept/concept_step_schedule.txt). 32^3 p3m farm job, CONCEPT's base steps. usage: rung_criterion_scan.py eta [eta ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_rungs import _setup, GOLD
from bachelor_thesis_n_body_amd import cosmo
sched = np.loadtxt(GOLD / "concept_step_schedule.txt")
idx = [104, 112, 120, 128, 136, 141]
print("NB_RUNG_SR_ONLY =", os.environ.get("NB_RUNG_SR_ONLY", "0"), "| logs: fraction above rung 0 at steps", idx, "=", sched[idx, 6], "| mean populations, steps 130-141:",
      np.round(sched[130:142, 8:12].mean(0), 1))
c, pos, vel = _setup(32, 12)
m = c.particle_mass(500.0, 32 ** 3)
for eta in [float(a) for a in sys.argv[1:]] or [0.018]:
    p, _, pops = cosmo.run_rungs(pos, vel, m, 500.0, c, 0.02, 1.0, levels=5, eta=eta, return_populations=True)
    fr = 1 - pops[:, 0] / 32 ** 3
    print(" eta", eta, np.round(fr[idx], 4), "| populations 130-141:", np.round(pops[130:142].mean(0), 1))
