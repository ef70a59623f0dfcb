"""Per-rank kernel work of the sink-decomposed N = 2^20 step at world sizes 1/2/4/8, emulated on ONE GPU (what each GPU of the driver's
scaling run executes per step, without the exchange), for both schedules of dist.ShardedEngine:
  one-sided : rank 0's P launches of (N/P sinks) x (N/P sources), accumulating (round 1/2)
  pairwise  : own chunk through the pair-symmetric kernel, the chunks of ranks -1 .. -(P-1)//2 two-sidedly (nb_accel_pair_dev: every
              A-B pair once; the visitor's partial sums would travel back), for an even P half of the pairs with the chunk half-way round (the other owner takes the other half)
Reports ms per step-equivalent and the interaction rate DELIVERED per GPU.   python tools/bench_shard_shapes.py [n] [f32|f64]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bachelor_thesis_n_body_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
tdt = torch.float64 if dtype == np.float64 else torch.float32
s = nb.plummer_sphere(n, 0, dtype)
pm = torch.from_numpy(np.vstack([s.pos.T, s.mass[None, :]])).contiguous().cuda()
k = nb.HipKernels(0)
for P in (1, 2, 4, 8):
    m = n // P
    H, half = (P - 1) // 2, P % 2 == 0
    chunks = [pm[:, r * m:(r + 1) * m].contiguous() for r in range(P)]
    acc = torch.zeros((3, m), dtype=tdt, device="cuda")
    out = torch.zeros((3, m), dtype=tdt, device="cuda")

    def one_sided():
        for h in range(P):
            k.accel(chunks[0], m, chunks[(0 - h) % P], m, 0.01, 1.0, acc, h > 0, True)

    def pairwise():
        k.accel(chunks[0], m, chunks[0], m, 0.01, 1.0, acc, False, True)          # own chunk: symmetric inside
        for h in range(1, H + 1):
            k.accel_pair(chunks[0], m, chunks[(0 - h) % P], m, 0.01, 1.0, acc, out, True, True)
        if half:          # the chunk half-way round: this rank's lower half x the whole visitor, two-sidedly (the other owner takes the other half)
            k.accel_pair(chunks[0][:, : m // 2], m // 2, chunks[P // 2], m, 0.01, 1.0, acc[:, : m // 2], out, True, True)

    for name, step in (("one-sided", one_sided), ("pairwise", pairwise)):
        if name == "one-sided":
            k.set_option("pair_symmetry", 0)
        else:
            k.set_option("pair_symmetry", -1)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"schedule": name, "world": P, "dtype": np.dtype(dtype).name, "particles_per_rank": m, "ms_per_step_per_rank": round(dt * 1e3, 3),
                          "interactions_per_s_per_gpu": float(m) * n / dt, "ideal_whole_job_interactions_per_s": float(n) * n / dt}), flush=True)
