"""Several realisations (seeds) of the reference's PP configuration: mean and scatter of P_here / P_ref per CONCEPT bin.
The reference table is ONE realisation, so the mean ratio over ours still carries the reference's own scatter
(~ 1/sqrt(modes) per bin) but not ours."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
ratios = []
for s in seeds:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_pp_reference.py"), "400", "64", str(s), "f32"],
                         capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in out.splitlines() if l and not l.startswith(("#", "{"))]
    ratios.append([float(r[5]) for r in rows])
    k = [float(r[0]) for r in rows]; modes = [int(r[2]) for r in rows]
r = np.array(ratios)
print("# k_ref  modes  mean(P_here/P_ref)  std over realisations   1/sqrt(modes)")
for i in range(r.shape[1]):
    print(f"{k[i]:.4e} {modes[i]:6d}   {r[:, i].mean():6.3f}   {r[:, i].std(ddof=1):6.3f}   {1 / np.sqrt(modes[i]):6.3f}")
print(json.dumps({"seeds": seeds, "mean_ratio_bins_6_16": float(r[:, 5:].mean()), "mean_ratio_bins_3_16": float(r[:, 2:].mean())}))
