import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb
from tools.explore import oracle_acc, relerr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
s = nb.plummer_sphere(n, 1, np.float64)
ref = oracle_acc(s.pos, s.mass, 0.01)
for ks in (1, 2, 4):
    for split in (0, 1, 8, 64):
        os.environ["NB_F64_KS"] = str(ks); os.environ["NB_F64_SPLIT"] = str(split)
        a = nb.compute_forces(s.pos, s.mass, 0.01)
        e = relerr(a, ref)
        bad = np.nonzero(e > 1e-8)[0]
        print(f"KS={ks} SPLIT={split}: max {e.max():.3e} median {np.median(e):.3e} nbad {bad.size} first bad {bad[:5]} last bad {bad[-5:]}", flush=True)
