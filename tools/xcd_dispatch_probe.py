"""Which XCD gets workgroup k of a launch — a fixed map, or does the dispatcher's round-robin pointer carry over from the launches before?
Launches the pair-symmetric kernel (NB_SYM_TRACE records every block's XCC_ID) after 0, 1, 2, 3, 5 one-block kernels on the same stream and
prints block index mod 8 -> XCC_ID each time.   python tools/xcd_dispatch_probe.py [n]"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/nb_sym_trace_probe.txt"
os.environ["NB_SYM_TRACE"] = path
import torch
import bachelor_thesis_n_body_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
s = nb.plummer_sphere(n, 0, np.float32)
k = nb.HipKernels(0)
pm = torch.from_numpy(np.vstack([s.pos.T, s.mass[None, :]])).contiguous().cuda()
acc = torch.zeros((3, n), dtype=torch.float32, device="cuda")
one = torch.zeros(1, device="cuda")
for extra in (0, 1, 2, 3, 5, 0):
    if os.path.exists(path):
        os.remove(path)
    k.accel(pm, n, pm, n, 0.01, 1.0, acc, False, uniform_mass=True)     # (the trace makes every launch synchronise)
    for _ in range(extra):
        one.add_(1.0)                                                    # a one-block kernel on the same stream
    os.remove(path)
    k.accel(pm, n, pm, n, 0.01, 1.0, acc, False, uniform_mass=True)
    torch.cuda.synchronize()
    rows = [[int(v) for v in line.split()] for line in open(path) if not line.startswith("#")]
    a = np.array(rows, dtype=np.int64)
    res = collections.Counter((int(r[0]) % 8, int(r[4])) for r in a)
    print(f"{extra} one-block kernels before the launch: block mod 8 -> XCC_ID", sorted(res.items()))
