"""Two ranks sharing the one GPU of the test box: the sink-decomposed engine with the REAL HIP kernels on torch device
memory, chunk-by-chunk accumulation and per-rank integration, the exchange carried by gloo (RCCL refuses two ranks on
one device). Everything except the RCCL transport itself is the code path of `bench.py --gpus N`."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, mode, dtype_name, out_dir, force_sym=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bachelor_thesis_n_body_amd as nb
        dtype = np.float32 if dtype_name == "f32" else np.float64
        torch.cuda.set_device(0)
        s = nb.plummer_sphere(n, 5, dtype)
        if dtype_name == "f64":
            s.mass = (s.mass * np.random.default_rng(1).uniform(0.5, 2.0, n)).astype(dtype)   # general-mass kernel
        eng = nb.ShardedEngine(n, dtype, device=torch.device("cuda", 0), mode=mode)
        if force_sym:                # chunks this small would take the one-sided kernels: force the pair-symmetric ones (own chunk and two-block hops)
            eng.k.set_option("sym_min_n", 0)
        eng.set_state(s)
        eng.accel(0.01)
        assert eng.pairwise == (world > 1) and (not force_sym or eng.own_chunk_symmetric)
        acc0 = eng.gather_state().acc
        eng.step(1 / 256, 3, 0.01)
        st = eng.gather_state()
        k, u = eng.energy(0.01)
        ks, us = eng.step_with_energy(1 / 256, 2, 0.01)       # fused force+potential kernels, phi accumulated over the chunks
        k2, u2 = eng.energy(0.01)                               # the separate ring pass at the same state
        if rank == 0:
            np.savez(Path(out_dir) / "res.npz", acc0=acc0, pos=st.pos, vel=st.vel, e=np.array([k, u]), ks=ks, us=us, e2=np.array([k2, u2]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,mode,dtype_name,force_sym", [(2, 6001, "ring", "f32", False), (2, 4096, "allgather", "f64", False), (4, 8191, "ring", "f32", False),
                                                               (3, 24001, "ring", "f32", True), (4, 20000, "ring", "f64", True), (5, 30011, "allgather", "f32", True)])
def test_two_ranks_on_one_gpu_match_the_single_engine(tmp_path, world, n, mode, dtype_name, force_sym):
    """2 to 5 ranks (uneven shards) sharing the one GPU of the box — at most 6 processes may use it. From 3 ranks on the pair-symmetric
    schedule has two-sided hops (nb_accel_pair_dev: the visiting chunk's partial sums travel back to its owner); force_sym runs them through
    the pair-symmetric kernels although the chunks are small."""
    import bachelor_thesis_n_body_amd as nb
    mp.spawn(_worker, args=(world, _free_port(), n, mode, dtype_name, str(tmp_path), force_sym), nprocs=world, join=True)
    r = np.load(tmp_path / "res.npz")
    dtype = np.float32 if dtype_name == "f32" else np.float64
    s = nb.plummer_sphere(n, 5, dtype)
    if dtype_name == "f64":
        s.mass = (s.mass * np.random.default_rng(1).uniform(0.5, 2.0, n)).astype(dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        acc0 = e.accel(0.01)
        e.step(1 / 256, 3, 0.01)
        ref = e.get_state()
        k, u = e.energy(0.01)
    tol = 3e-6 if dtype_name == "f32" else 1e-12          # chunked accumulation: different (fixed) summation order
    scale = np.abs(acc0).max()
    assert np.abs(r["acc0"] - acc0).max() < tol * scale
    assert np.abs(r["pos"] - ref.pos).max() < tol * 10 and np.abs(r["vel"] - ref.vel).max() < tol * 10
    assert r["e"][0] == pytest.approx(k, rel=1e-6) and r["e"][1] == pytest.approx(u, rel=1e-6)
    # sharded step_with_energy: last logged energies = the separate potential/kinetic ring pass; whole log = the single engine's
    etol = 3e-6 if dtype_name == "f32" else 1e-11
    assert r["ks"][-1] == pytest.approx(r["e2"][0], rel=1e-12) and r["us"][-1] == pytest.approx(r["e2"][1], rel=etol)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        e.step(1 / 256, 3, 0.01)
        K, U = e.step_with_energy(1 / 256, 2, 0.01)
    np.testing.assert_allclose(r["ks"], K, rtol=1e-5 if dtype_name == "f32" else 1e-11)
    np.testing.assert_allclose(r["us"], U, rtol=etol)
