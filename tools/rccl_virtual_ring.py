"""Rank 0's share of the multi-GPU headline run, with the REAL RCCL transport, on one GPU: `tools.virtual_ring.VirtualRingEngine` — the one real rank plays
rank 0 of a P-rank ring and passes the other ranks' source blocks through the communicator to itself (tests/test_gpu_rccl.py). Prints, for P = 1, 2, 4, 8 at
N = 2^20 fp32 (BASELINE configs[2] shapes): time per force evaluation of rank 0's N/P sinks against all N sources, per-hop exchange time, the kernel it overlaps
and the part of the exchange the compute stream had to wait for. What this cannot show is the xGMI link itself (a self send is a device-local copy).
usage: rccl_virtual_ring.py [n] [f32|f64] [pairwise]
"pairwise": the pair-symmetric schedule (round 3): rank 0 works two-sidedly on the chunks of ranks -1 .. -(P-1)//2 and sends the visitors' partial sums through the
communicator; what the ranks holding ITS chunk would return is computed locally in this one-rank rehearsal (extra kernels that a real rank does not run), so the
per-evaluation wall time over-counts — `rank0_kernel_ms` (HIP events around the kernels a real rank 0 runs) is the figure to read."""
import json, os, sys, time
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bachelor_thesis_n_body_amd as nb
from bachelor_thesis_n_body_amd.dist import HopProbe
from tools.virtual_ring import VirtualRingEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
pairwise = len(sys.argv) > 3 and sys.argv[3] == "pairwise"
hp = os.environ.get("NB_RCCL_HIGH_PRIORITY", "1") != "0"


class Timer:
    def __init__(self):
        self.pairs, self.enabled = [], False

    def begin(self):
        if self.enabled:
            self._e = torch.cuda.Event(enable_timing=True); self._e.record()

    def end(self, interactions):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True); e.record(); self.pairs.append((self._e, e))

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs)

opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=hp)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
print(json.dumps({"communicator_stream_high_priority": hp}), flush=True)
s = nb.plummer_sphere(n, 0, dtype)
for P in (1, 2, 4, 8):
    eng = VirtualRingEngine(n, dtype, device=dev, virtual_world=P, pairwise=pairwise) if P > 1 else nb.ShardedEngine(n, dtype, device=dev)
    eng.set_state(s)
    probe = HopProbe(torch, dev); eng.hop_probe = probe
    tm = Timer(); eng.kernel_timer = tm
    for _ in range(2):
        eng.accel(0.01)
    torch.cuda.synchronize()
    probe.enabled = P > 1
    tm.enabled = True
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.accel(0.01)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ps = probe.summary() or {}
    print(json.dumps({"virtual_ranks": P, "schedule": "pair-symmetric" if (pairwise and P > 1) else "one-sided", "rank0_kernel_ms": tm.total_ms() / reps, "n": n, "dtype": np.dtype(dtype).name, "sinks_of_rank0": eng.n_loc, "ms_per_force_evaluation": dt * 1e3,
                      "interactions_per_s_of_this_rank": eng.n_loc * float(n) / dt, "projected_whole_job_interactions_per_s": P * eng.n_loc * float(n) / dt,
                      "hop_comm_ms": ps.get("comm_ms"), "hop_kernel_ms": ps.get("kernel_ms"), "hop_exposed_ms": ps.get("exposed_ms"),
                      "back_comm_ms": ps.get("back_comm_ms"), "back_exposed_ms": ps.get("back_exposed_ms"), "add_ms": ps.get("add_ms")}), flush=True)
dist.destroy_process_group()
