"""Per-rank kernel work of the sink-decomposed N = 2^20 fp32 step at world sizes 1/2/4/8, emulated on ONE GPU: rank 0's
P launches of (N/P sinks) x (N/P sources), accumulating — what each GPU of the driver's scaling run executes per step
(without the exchange). Reports ms per step-equivalent and the interaction rate of one GPU at that shape."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bachelor_thesis_n_body_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
s = nb.plummer_sphere(n, 0, np.float32)
pm = torch.from_numpy(np.vstack([s.pos.T, s.mass[None, :]])).contiguous().cuda()
k = nb.HipKernels(0)
for P in (1, 2, 4, 8):
    m = n // P
    snk = pm[:, :m].contiguous()
    chunks = [pm[:, r * m:(r + 1) * m].contiguous() for r in range(P)]
    acc = torch.zeros((3, m), dtype=torch.float32, device="cuda")
    def step():
        for h in range(P):
            k.accel(snk, m, chunks[(0 - h) % P], m, 0.01, 1.0, acc, h > 0, True)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"world": P, "sinks_per_rank": m, "launches_per_step": P, "ms_per_step_per_rank": dt * 1e3,
                      "interactions_per_s_per_gpu": float(m) * n / dt, "ideal_whole_job_interactions_per_s": float(n) * n / dt}), flush=True)
