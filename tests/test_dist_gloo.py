"""World-size-2 (and 3, uneven) rehearsal of the sink-decomposed path on CPU with the gloo backend: the ring
exchange, partitioning, accumulate-over-chunks and the energy all-reduce are the product code; only the
arithmetic is stubbed by the oracle (tests/cpu_kernels.py). Results must match the single-process oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, mode, pairwise, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bachelor_thesis_n_body_amd as nb
        from cpu_kernels import OracleKernels
        eps, dt = 0.02, 1 / 128
        s = nb.plummer_sphere(n, 5)
        s.mass = s.mass * np.random.default_rng(1).uniform(0.5, 2.0, n)   # unequal masses
        eng = nb.ShardedEngine(n, np.float64, device="cpu", kernels=OracleKernels(), mode=mode, pairwise=pairwise)
        assert sum(eng.counts) == n and eng.n_loc == eng.counts[rank] and eng.pairwise == pairwise
        eng.set_state(s)
        from bachelor_thesis_n_body_amd.dist import HopProbe
        probe = HopProbe(torch, torch.device("cpu"))       # per-hop exchange timing (bench.py's `ring` block) on the host clock
        eng.hop_probe = probe
        probe.enabled = mode == "ring"
        eng.accel(eps)
        probe.enabled = False
        if mode == "ring":
            ps = probe.summary()
            # position hops: all P - 1 other chunks one-sidedly, or P // 2 with the pair-symmetric schedule (the other chunks' owners do the work)
            assert ps["hops_timed"] == (world // 2 if pairwise else world - 1) and ps["comm_ms"] >= 0.0
            if pairwise:     # the return leg: one partial per two-sided hop ((P - 1) // 2, + the split half-way hop of an even ring) and the closing adds
                assert ps["returns_timed"] == world // 2 and ps["evaluations_timed"] == 1 and ps["back_comm_ms"] >= 0 and ps["add_ms"] >= 0
            else:
                assert "back_comm_ms" not in ps
        else:
            assert probe.summary() is None
        acc0 = eng.gather_state().acc
        k0, u0 = eng.energy(eps)
        eng.step(dt, 3, eps)
        st = eng.gather_state()
        k1, u1 = eng.energy(eps)
        ks, us = eng.step_with_energy(dt, 2, eps)            # fused potential: energies of steps 4 and 5
        k2, u2 = eng.energy(eps)
        if rank == 0:
            np.savez(Path(out_dir) / "res.npz", acc0=acc0, pos=st.pos, vel=st.vel, mass=st.mass, acc=st.acc,
                     e=np.array([k0, u0, k1, u1]), fused=np.array([ks[-1], us[-1], k2, u2]), ks=ks, us=us)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,mode,pairwise", [(2, 96, "ring", True), (2, 97, "allgather", True), (3, 50, "ring", True), (4, 131, "ring", True),
                                                   (5, 77, "ring", True), (6, 80, "ring", True), (4, 61, "allgather", True), (5, 49, "allgather", True),
                                                   (7, 45, "ring", True), (8, 43, "ring", True),
                                                   (2, 96, "ring", False), (3, 50, "ring", False), (4, 131, "allgather", False)])
def test_sharded_engine_matches_single_process_oracle(tmp_path, world, n, mode, pairwise):
    """pairwise: every pair of chunks is worked on by one of its owners (two-sided kernel entry, partial sums sent back point to point; the
    chunk half-way round an even ring split between its two owners) — worlds 3..8 cover one to three returned partials, the re-use of the two
    partial buffers in flight and the half-way hop (8 ranks: the shape of the scaling run, chunks of 5-6 particles); pairwise=False: every rank sums all chunks for its own sinks."""
    from oracle import nbody_oracle as ora
    import bachelor_thesis_n_body_amd as nb
    mp.spawn(_worker, args=(world, _free_port(), n, mode, pairwise, str(tmp_path)), nprocs=world, join=True)
    r = np.load(tmp_path / "res.npz")
    eps, dt = 0.02, 1 / 128
    s = nb.plummer_sphere(n, 5)
    mass = s.mass * np.random.default_rng(1).uniform(0.5, 2.0, n)
    o = ora.State(s.pos.copy(), s.vel.copy(), mass)
    np.testing.assert_allclose(r["acc0"], ora.compute_forces(o.pos, o.mass, eps), rtol=1e-12, atol=1e-14)
    k0, u0 = ora.energy(o, eps)
    ora.step(o, dt, 3, eps=eps)
    k1, u1 = ora.energy(o, eps)
    np.testing.assert_allclose(r["pos"], o.pos, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(r["vel"], o.vel, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(r["acc"], o.acc, rtol=1e-11, atol=1e-13)
    np.testing.assert_array_equal(r["mass"], mass)
    np.testing.assert_allclose(r["e"], [k0, u0, k1, u1], rtol=1e-12)
    # step_with_energy: per-step energies of the whole system, equal to the separate ring pass and to the oracle
    np.testing.assert_allclose(r["fused"][:2], r["fused"][2:], rtol=1e-12)
    for i in range(2):
        ora.step(o, dt, 1, eps=eps)
        np.testing.assert_allclose([r["ks"][i], r["us"][i]], ora.energy(o, eps), rtol=1e-11)


def test_single_rank_engine_without_process_group():
    """world = 1 needs no torch.distributed initialisation (bench.py --gpus 1 path)."""
    import bachelor_thesis_n_body_amd as nb
    from cpu_kernels import OracleKernels
    from oracle import nbody_oracle as ora
    s = nb.plummer_sphere(40, 2)
    eng = nb.ShardedEngine(40, np.float64, device="cpu", kernels=OracleKernels())
    eng.set_state(s)
    eng.step(0.01, 2, 0.05)
    o = ora.State(s.pos.copy(), s.vel.copy(), s.mass.copy())
    ora.step(o, 0.01, 2, eps=0.05)
    np.testing.assert_allclose(eng.gather_state().pos, o.pos, rtol=1e-13, atol=1e-15)
    with pytest.raises(RuntimeError, match="HIP device"):
        nb.ShardedEngine(40, np.float64, device="cpu")          # no silent CPU path in the product


def _switch_worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bachelor_thesis_n_body_amd as nb
        from cpu_kernels import OracleKernels
        s = nb.plummer_sphere(n, 9)
        # the ranks DISAGREE on the flag (as with NB_PAIRWISE_SCHEDULE exported on some nodes only): the engine takes the MIN over the group
        eng = nb.ShardedEngine(n, np.float64, device="cpu", kernels=OracleKernels(), pairwise=(rank != 1))
        assert eng.pairwise is False and eng.H == 0
        eng.set_state(s)
        beats = []
        eng.heartbeat = beats.append
        eng.accel(0.02)
        a_one = eng.gather_state().acc
        eng.set_schedule(True)                      # live switch (bench.py's A/B): same state, the other message pattern
        assert eng.pairwise and eng.H == (world - 1) // 2 and eng.split_half == (world % 2 == 0)
        eng.accel(0.02)
        a_pair = eng.gather_state().acc
        assert len(beats) == world // 2 + 1 and beats[0] == "hop 0"
        eng.set_schedule(False)
        eng.step(1 / 128, 2, 0.02)
        eng.set_schedule(True)
        eng.step(1 / 128, 2, 0.02)
        st = eng.gather_state()
        if rank == 0:
            np.savez(Path(out_dir) / "res.npz", a_one=a_one, a_pair=a_pair, pos=st.pos)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(3, 70), (4, 90)])
def test_schedule_flag_agreement_and_live_switch(tmp_path, world, n):
    from oracle import nbody_oracle as ora
    import bachelor_thesis_n_body_amd as nb
    mp.spawn(_switch_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    r = np.load(tmp_path / "res.npz")
    s = nb.plummer_sphere(n, 9)
    ref = ora.compute_forces(s.pos, s.mass, 0.02)
    np.testing.assert_allclose(r["a_one"], ref, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(r["a_pair"], ref, rtol=1e-12, atol=1e-14)
    o = ora.State(s.pos.copy(), s.vel.copy(), s.mass.copy())
    ora.step(o, 1 / 128, 4, eps=0.02)
    np.testing.assert_allclose(r["pos"], o.pos, rtol=1e-12, atol=1e-14)
