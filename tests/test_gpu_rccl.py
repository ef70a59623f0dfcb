"""The RCCL branch of the sharded engine on the one-GPU test box. RCCL refuses two ranks on one device, but a single rank may send to and
receive from itself: `tools.virtual_ring.VirtualRingEngine` (a ShardedEngine with its two transport hooks replaced) lets the one real rank play
rank 0 of a P-rank ring, holding the other ranks' source blocks locally and passing them through the communicator hop by hop. That runs — with the real transport — exactly what the gloo rehearsals
cannot: the `nccl` branch of `_for_each_chunk` (no host fence: the exchange is posted before the kernel and ordered by the communicator's
stream), receive-buffer reuse under that ordering, `wait()`, and the HIP-event path of `HopProbe`."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CODE = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    import bachelor_thesis_n_body_amd as nb
    from bachelor_thesis_n_body_amd.dist import HopProbe
    from tools.virtual_ring import VirtualRingEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # pairwise: the pair-symmetric schedule (two-sided hops, the visiting chunk's partial sums sent back — here to itself — and the returned
        # partials added at the end); what the ranks holding rank 0's chunk would return is computed locally in this one-rank rehearsal.
        # Even P: the chunk half-way round is SPLIT between its two owners (column-sliced views into the two-block kernel, out/back swap with
        # the same peer) — the shape of the 2/4/8-GPU scaling run, here under the communicator's ordering with no host fence
        for dtype, P, n, general, pairwise in ((np.float32, 2, 50001, False, False), (np.float32, 4, 131072, False, False), (np.float64, 3, 20011, True, False),
                                               (np.float32, 8, 65536, True, False), (np.float32, 3, 150001, False, True), (np.float32, 8, 262144, False, True),
                                               (np.float64, 5, 100003, True, True), (np.float32, 4, 40001, True, True), (np.float32, 2, 131071, False, True),
                                               (np.float64, 6, 60007, False, True)):
            s = nb.plummer_sphere(n, 7, dtype)
            if general:
                s.mass = (s.mass * np.random.default_rng(2).uniform(0.5, 2.0, n)).astype(dtype)
            with nb.Engine(n, dtype) as e:
                e.set_state(s)
                ref = e.accel(0.01)
            eng = VirtualRingEngine(n, dtype, device=torch.device("cuda", 0), virtual_world=P, pairwise=pairwise)
            assert dist.get_backend() == "nccl" and eng.world == P and eng.n_loc == eng.counts[0] and eng.pairwise == pairwise
            assert eng.split_half == (pairwise and P %% 2 == 0)
            eng.set_state(s)
            probe = HopProbe(torch, torch.device("cuda", 0)); eng.hop_probe = probe; probe.enabled = True
            for rep in range(4):                      # repeated evaluations: every receive buffer is reused under the communicator's ordering
                eng.accel(0.01)
            torch.cuda.synchronize()
            got = eng.local_acc()
            tol = 3e-6 if dtype == np.float32 else 1e-12
            err = np.abs(got - ref[: eng.n_loc]).max() / np.abs(ref).max()
            assert err < tol, (dtype, P, err)
            ps = probe.summary()
            assert ps["hops_timed"] == 4 * (P // 2 if pairwise else P - 1) and ps["comm_ms"] > 0 and ps["kernel_ms"] > 0 and ps["exposed_ms"] >= 0
            if pairwise:      # the return leg is timed too: one partial per two-sided hop (P // 2 for an even ring: the split hop returns one as well)
                assert ps["returns_timed"] == 4 * (P // 2) and ps["back_comm_ms"] > 0 and ps["back_exposed_ms"] >= 0 and ps["add_ms"] > 0 and ps["evaluations_timed"] == 4
                # fused potential through the same schedule: phi of rank 0's sinks against the one-sided potential-carrying kernel on all sources
                eng.hop_probe = None
                eng.accel(0.01, with_potential=True)
                torch.cuda.synchronize()
                t_dtype = torch.float32 if dtype == np.float32 else torch.float64
                allsrc = torch.from_numpy(np.ascontiguousarray(np.vstack([s.pos.T, s.mass[None, :]]))).to("cuda:0", t_dtype)
                a2 = torch.zeros(3, eng.n_loc, dtype=t_dtype, device="cuda:0"); phi2 = torch.zeros(eng.n_loc, dtype=torch.float64, device="cuda:0")
                eng.k.set_option("pair_symmetry", 0)
                eng.k.accel_pot(eng.pm, eng.n_loc, allsrc, n, 0.01, 1.0, a2, phi2, False, not general)
                eng.k.set_option("pair_symmetry", -1)
                torch.cuda.synchronize()
                perr = float(((eng.phi[: eng.n_loc] - phi2).abs().max() / phi2.abs().max()).item())
                assert perr < (3e-6 if dtype == np.float32 else 1e-12), (dtype, P, perr)
            print("ok", np.dtype(dtype).name, P, n, "pairwise" if pairwise else "one-sided", "err %%.2e comm_ms %%.3f kernel_ms %%.3f exposed_ms %%.3f" %% (err, ps["comm_ms"], ps["kernel_ms"], ps["exposed_ms"]))
    finally:
        dist.destroy_process_group()
""")


def test_rccl_ring_with_one_real_rank_playing_rank_zero_of_p():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CODE % (str(ROOT), port)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.stdout.count("ok ") == 10, out.stdout
    print(out.stdout)
