"""Sink-index decomposition over the GPUs of one node: one process per GPU under
torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" for the CPU rehearsal in tests/).

Rank g of P owns the contiguous sink range [off_g, off_g + cnt_g) and integrates only those particles. Two schedules
move the (x, y, z, m) blocks between the ranks:

* one-sided (`pairwise=False`; the schedule SURVEY.md §8(e) / north_star name): every rank needs ALL sources, so each
  force evaluation circulates the block of every rank once around a ring of P-1 hops: while the kernel works on the
  chunk that arrived at hop h, the chunk for hop h+1 is already in flight on the communicator's stream (send to
  rank+1 / receive from rank-1, posted BEFORE the kernel launch so the collective does not queue behind it).
  Accelerations need no reduction — each sink's sum completes locally.
* pair-symmetric (`pairwise=True`, the default for P > 1): every pair of chunks is worked on by ONE of its two
  owners with the two-block kernel (each A-B pair evaluated once), so the position ring runs only P//2 hops — and
  the visiting chunk's PARTIAL ACCELERATIONS TRAVEL BACK to its owner point to point (12 B per particle and hop,
  posted while the next kernel runs) and are added in a fixed order at the end. Half the pair work per rank for one
  extra message per hop; `ShardedEngine.set_schedule()` switches between the two on a live engine (bench.py times
  both in one run).

Only the two energy scalars are all-reduced. One position message = 16*cnt bytes (fp32) / 32*cnt bytes (fp64), i.e.
2 MiB per hop at N = 2^20 on 8 GPUs versus ~2-3 ms of math per chunk: the ring is hidden behind compute.

torch is used for device memory, streams and the process group only; all arithmetic is done by the HIP kernels
behind the C-ABI (`HipKernels`). The compute backend is injectable so that the ring schedule, partitioning and
buffer rotation can be exercised on CPU tensors with gloo (tests/ pass a NumPy stand-in; nothing in this package
imports the oracle). The transport of one hop sits in two overridable hooks (`_post_hop`, `_post_return`):
tools/virtual_ring.py replaces them to rehearse a P-rank ring with ONE real RCCL rank.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from . import _lib
from .state import State


def partition(n_total: int, world: int) -> Tuple[List[int], List[int]]:
    """Contiguous, balanced sink ranges: the first n_total % world ranks own one extra particle."""
    base, rem = divmod(int(n_total), int(world))
    counts = [base + (1 if r < rem else 0) for r in range(world)]
    offsets = [0] * world
    for r in range(1, world):
        offsets[r] = offsets[r - 1] + counts[r - 1]
    return counts, offsets


class HipKernels:
    """Device-pointer entry points of libnbody_amd.so applied to torch CUDA(=HIP) tensors on the current stream."""

    def __init__(self, device_index: int):
        import torch

        self._torch = torch
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        _lib.check(None, self._lib.nb_create(C.byref(self._ctx), 1, _lib.NB_F32, device_index), "nb_create")
        self._stream = self._stream_obj = None

    def close(self):
        if self._ctx:
            self._lib.nb_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int) -> None:
        """nb_set_option on the launch context: 'pair_symmetry' 0/1, 'sym_min_n', 'sym_slab_bytes' (value < 0: default)."""
        _lib.check(self._ctx, self._lib.nb_set_option(self._ctx, _lib.OPTIONS[name], int(value)),
                   "nb_set_option")

    def last_accel_symmetric(self) -> bool:
        return bool(self._lib.nb_last_accel_symmetric(self._ctx))

    def _bind(self):
        cur = self._torch.cuda.current_stream()
        s = cur.cuda_stream
        if s != self._stream:
            _lib.check(self._ctx, self._lib.nb_set_stream(self._ctx, C.c_void_p(s)), "nb_set_stream")
            # hold the Stream OBJECT, not only its handle: the context keeps using (and synchronising) the handle until the
            # next nb_set_stream / nb_destroy, so it must not be destroyed under it (include/nbody_amd.h: nb_set_stream)
            self._stream, self._stream_obj = s, cur

    @staticmethod
    def _code(t) -> int:
        import torch

        if t.dtype == torch.float32:
            return _lib.NB_F32
        if t.dtype == torch.float64:
            return _lib.NB_F64
        raise TypeError(f"unsupported tensor dtype {t.dtype}")

    @staticmethod
    def _rows(t, k):
        """Device pointers of the first k rows of a [rows][n] block; a column range of such a block (rows still unit-stride) is fine too."""
        assert t.dim() == 2 and t.shape[0] >= k and (t.shape[1] <= 1 or t.stride(1) == 1)
        es, base = t.element_size(), t.data_ptr()
        return [C.c_void_p(base + i * t.stride(0) * es) for i in range(k)]

    def accel(self, snk, n_snk, src, n_src, eps, G, acc, accumulate, uniform_mass=False):
        self._bind()
        xi, yi, zi = self._rows(snk, 3)
        xs, ys, zs, ms = self._rows(src, 4)
        ax, ay, az = self._rows(acc, 3)
        rc = self._lib.nb_accel_dev(self._ctx, self._code(snk), xi, yi, zi, n_snk, xs, ys, zs, ms, n_src, eps, G,
                                    ax, ay, az, (_lib.NB_ACCUMULATE if accumulate else 0) | (_lib.NB_UNIFORM_MASS if uniform_mass else 0))
        _lib.check(self._ctx, rc, "nb_accel_dev")

    def accel_kdk(self, snk, n_snk, src, n_src, eps, G, acc, accumulate, vel, kick_close, kick_open=0.0, drift=0.0, uniform_mass=False):
        """accel() followed by the leapfrog tail on the sinks (nb_accel_kdk_dev): vel += acc*kick_close and, when kick_open or drift is
        non-zero, vel += acc*kick_open; snk[:3] += vel*drift — applied inside the force's last kernel where the pair-symmetric kernels run."""
        self._bind()
        xi, yi, zi = self._rows(snk, 3)
        xs, ys, zs, ms = self._rows(src, 4)
        ax, ay, az = self._rows(acc, 3)
        vx, vy, vz = self._rows(vel, 3)
        rc = self._lib.nb_accel_kdk_dev(self._ctx, self._code(snk), xi, yi, zi, n_snk, xs, ys, zs, ms, n_src, eps, G, ax, ay, az,
                                        (_lib.NB_ACCUMULATE if accumulate else 0) | (_lib.NB_UNIFORM_MASS if uniform_mass else 0),
                                        vx, vy, vz, kick_close, kick_open, drift)
        _lib.check(self._ctx, rc, "nb_accel_kdk_dev")

    def kick_drift(self, pm, vel, acc, n, kick, drift):
        self._bind()
        x, y, z = self._rows(pm, 3)
        vx, vy, vz = self._rows(vel, 3)
        ax, ay, az = self._rows(acc, 3)
        _lib.check(self._ctx, self._lib.nb_kick_drift_dev(self._ctx, self._code(vel), x, y, z, vx, vy, vz, ax, ay, az, n, kick, drift), "nb_kick_drift_dev")

    def accel_pot(self, snk, n_snk, src, n_src, eps, G, acc, phi, accumulate, uniform_mass=False):
        """accel() that also fills phi (float64[n_snk]): sum_j m_j / sqrt(r^2 + eps^2), self term included, no G."""
        self._bind()
        assert phi.dtype == self._torch.float64 and phi.is_contiguous() and phi.numel() >= n_snk
        xi, yi, zi = self._rows(snk, 3)
        xs, ys, zs, ms = self._rows(src, 4)
        ax, ay, az = self._rows(acc, 3)
        rc = self._lib.nb_accel_pot_dev(self._ctx, self._code(snk), xi, yi, zi, n_snk, xs, ys, zs, ms, n_src, eps, G,
                                        ax, ay, az, C.c_void_p(phi.data_ptr()),
                                        (_lib.NB_ACCUMULATE if accumulate else 0) | (_lib.NB_UNIFORM_MASS if uniform_mass else 0))
        _lib.check(self._ctx, rc, "nb_accel_pot_dev")

    def accel_pair(self, blk_a, n_a, blk_b, n_b, eps, G, acc_a, acc_b, accumulate, uniform_mass=False, phi_a=None, phi_b=None):
        """Two different blocks, every A-B pair once (nb_accel_pair_dev): acc_a (+)= force of B on A, acc_b = force of A on B (overwritten);
        phi_a / phi_b (float64, both or neither): the fused potential, phi_a accumulating like acc_a."""
        self._bind()
        xa, ya, za, ma = self._rows(blk_a, 4)
        xb, yb, zb, mb = self._rows(blk_b, 4)
        aax, aay, aaz = self._rows(acc_a, 3)
        bax, bay, baz = self._rows(acc_b, 3)
        pa = C.c_void_p(phi_a.data_ptr()) if phi_a is not None else None
        pb = C.c_void_p(phi_b.data_ptr()) if phi_b is not None else None
        rc = self._lib.nb_accel_pair_dev(self._ctx, self._code(blk_a), xa, ya, za, ma, n_a, xb, yb, zb, mb, n_b, eps, G, aax, aay, aaz,
                                         bax, bay, baz, pa, pb,
                                         (_lib.NB_ACCUMULATE if accumulate else 0) | (_lib.NB_UNIFORM_MASS if uniform_mass else 0))
        _lib.check(self._ctx, rc, "nb_accel_pair_dev")

    def pot_energy(self, phi, mass_row, n, eps, out, uniform_mass=False):
        """out[0] = sum_i m_i (phi_i - self term) for the phi of accel_pot() accumulated over all source chunks."""
        self._bind()
        rc = self._lib.nb_pot_energy_dev(self._ctx, self._code(mass_row), C.c_void_p(phi.data_ptr()), C.c_void_p(mass_row.data_ptr()), n,
                                         eps, int(bool(uniform_mass)), C.c_void_p(out.data_ptr()))
        _lib.check(self._ctx, rc, "nb_pot_energy_dev")

    def potential(self, snk, n_snk, snk_g0, src, n_src, src_g0, eps, out, accumulate):
        self._bind()
        xi, yi, zi, mi = self._rows(snk, 4)
        xs, ys, zs, ms = self._rows(src, 4)
        rc = self._lib.nb_potential_dev(self._ctx, self._code(snk), xi, yi, zi, mi, n_snk, snk_g0, xs, ys, zs, ms,
                                        n_src, src_g0, eps, C.c_void_p(out.data_ptr()), int(bool(accumulate)))
        _lib.check(self._ctx, rc, "nb_potential_dev")

    def kinetic(self, vel, mass_row, n, out):
        self._bind()
        vx, vy, vz = self._rows(vel, 3)
        rc = self._lib.nb_kinetic_dev(self._ctx, self._code(vel), vx, vy, vz, C.c_void_p(mass_row.data_ptr()), n,
                                      C.c_void_p(out.data_ptr()))
        _lib.check(self._ctx, rc, "nb_kinetic_dev")

    def kick(self, vel, acc, n, h):
        self._bind()
        vx, vy, vz = self._rows(vel, 3)
        ax, ay, az = self._rows(acc, 3)
        _lib.check(self._ctx, self._lib.nb_kick_dev(self._ctx, self._code(vel), vx, vy, vz, ax, ay, az, n, h), "nb_kick_dev")

    def drift(self, pm, vel, n, dt):
        self._bind()
        x, y, z = self._rows(pm, 3)
        vx, vy, vz = self._rows(vel, 3)
        _lib.check(self._ctx, self._lib.nb_drift_dev(self._ctx, self._code(vel), x, y, z, vx, vy, vz, n, dt), "nb_drift_dev")


class HopProbe:
    """Per-hop timing of the ring: how long each exchange took (`comm_ms`, from the moment it was posted until its data
    had landed) next to the kernel it was posted beside (`kernel_ms`), and how long the compute stream then still had
    to wait for it (`exposed_ms` — 0 when the hop is hidden behind the math). The pair-symmetric schedule's return leg is
    timed the same way: `back_comm_ms` (a partial-sum message from post to arrival), `back_exposed_ms` (what the compute
    stream waited for returned partials, per force evaluation) and `add_ms` (the closing `acc += back[h]` adds, per force
    evaluation) — so a gap between `kernel_ms_per_step` and `ms_per_step` can be attributed. Device runs use HIP events:
    the post and the kernel are bracketed on the compute stream; the arrival is observed on a side stream that waits for the
    communicator's work only, so it does not include the kernel. Host-blocking transports (gloo) and CPU rehearsals
    fall back to the host clock around `wait()`."""

    def __init__(self, torch, device):
        import time
        self._t, self._clock = torch, time.perf_counter
        self.cuda = device.type == "cuda"
        self.side = torch.cuda.Stream(device) if self.cuda else None
        self.enabled = False
        self.hops = []      # per hop: (post, comm_done, kernel_begin, kernel_end, after_wait)
        self.backs = []     # per returned partial: (post, arrived)
        self.back_waits = []   # (before, after) of every wait for a returned partial on the compute stream
        self.closings = []  # per force evaluation: (adds_begin, adds_end)
        self.evals = 0

    def _mark(self, stream=None):
        if not self.cuda:
            return self._clock()
        e = self._t.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        return e

    def exchange(self, reqs, launch, host_blocking):
        """Run `launch()` (the kernel on the chunk in hand) beside the posted requests `reqs` and time both."""
        post = self._hold
        k0 = self._mark()
        launch()
        k1 = self._mark()
        if self.cuda and not host_blocking:
            with self._t.cuda.stream(self.side):     # the side stream waits for the communicator only
                for q in reqs:
                    q.wait()
                done = self._mark(self.side)
            for q in reqs:
                q.wait()                             # ... the compute stream as usual
            self.hops.append((post, done, k0, k1, self._mark()))
        else:
            for q in reqs:
                q.wait()
            t = self._clock()
            self.hops.append((self._host_post, t, k0, k1, self._mark()))

    def posting(self):
        self._hold = self._mark()
        self._host_post = self._clock()

    def returning(self, post_mark, reqs, host_blocking):
        """A partial-sum message has just been posted (`post_mark` = posting mark taken before it): observe its arrival on the side stream."""
        if self.cuda and not host_blocking:
            with self._t.cuda.stream(self.side):
                for q in reqs:
                    q.wait()
                self.backs.append((post_mark, self._mark(self.side)))
        else:
            self._pending_back = getattr(self, "_pending_back", {})
            self._pending_back[id(reqs)] = self._clock()

    def wait_back(self, reqs):
        """Wait for a returned partial on the compute stream (host for blocking transports) and record what the wait cost."""
        w0 = self._mark()
        for q in reqs:
            q.wait()
        w1 = self._mark()
        self.back_waits.append((w0, w1))
        t0 = getattr(self, "_pending_back", {}).pop(id(reqs), None)
        if t0 is not None:
            self.backs.append((t0, self._clock()))

    def closing(self, fn):
        """Run the closing adds of one force evaluation between two marks."""
        a0 = self._mark()
        fn()
        self.closings.append((a0, self._mark()))
        self.evals += 1

    def summary(self):
        """Means over the recorded hops (call after a device synchronise)."""
        if not self.hops:
            return None
        ms = lambda a, b: a.elapsed_time(b) if self.cuda and not isinstance(a, float) else (b - a) * 1e3
        comm = [ms(p, d) for p, d, _, _, _ in self.hops]
        kern = [ms(k0, k1) for _, _, k0, k1, _ in self.hops] if self.cuda else [float("nan")] * len(self.hops)
        expo = [max(0.0, ms(k1, a)) for _, _, _, k1, a in self.hops] if self.cuda else [float("nan")] * len(self.hops)
        mean = lambda v: sum(v) / len(v)
        out = {"hops_timed": len(self.hops), "comm_ms": mean(comm), "kernel_ms": mean(kern), "exposed_ms": mean(expo),
               "comm_ms_max": max(comm)}
        if self.backs or self.closings:
            back = [ms(p, d) for p, d in self.backs]
            ev = max(1, self.evals)
            out.update({"returns_timed": len(back), "back_comm_ms": mean(back) if back else 0.0, "back_comm_ms_max": max(back) if back else 0.0,
                        # per force evaluation: what the compute stream (host, for a blocking transport) waited for returned partials, and the closing adds
                        "back_exposed_ms": sum(max(0.0, ms(a, b)) for a, b in self.back_waits) / ev,
                        "add_ms": sum(ms(a, b) for a, b in self.closings) / ev, "evaluations_timed": self.evals})
        return out


class ShardedEngine:
    """Sink-decomposed engine of one rank. All ranks must make the same sequence of calls."""

    def __init__(self, n_total: int, dtype=np.float32, device=None, group=None, kernels=None, mode: str = "ring", pairwise=None, world=None):
        """pairwise (default: on for more than one rank when the kernel backend offers accel_pair; NB_PAIRWISE_SCHEDULE=0 in the environment
        turns the default off; the ranks agree on the flag with a MIN all-reduce, so a rank that sees another environment cannot post a
        different message sequence): every pair of chunks is worked on by ONE of its two owners — rank r takes the chunks of ranks
        r-1 .. r-(P-1)//2 two-sidedly (nb_accel_pair_dev: each A-B pair evaluated once) and sends the other owner its partial sums, point to
        point; for an even P the chunk half-way round is split between its two owners (the lower rank takes its own lower half against the
        whole visitor, the higher rank its whole chunk against the visitor's upper half; they swap partials). The position ring then runs
        P//2 hops instead of P-1 and the pair work per rank halves. pairwise=False: every rank sums all P chunks
        one-sidedly for its own sinks (no accelerations travel). `world` is for subclasses that play several ranks with one real one
        (tools/virtual_ring.py)."""
        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = int(world) if world else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.n_total = int(n_total)
        self.counts, self.offsets = partition(self.n_total, self.world)
        self.n_loc, self.off = self.counts[self.rank], self.offsets[self.rank]
        self.n_max = max(self.counts)
        if mode not in ("ring", "allgather"):
            raise ValueError("mode must be 'ring' or 'allgather'")
        self.mode = mode
        self.np_dtype = np.dtype(dtype)
        self.t_dtype = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[self.np_dtype]
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if kernels is None:
            if self.device.type != "cuda":
                raise RuntimeError("ShardedEngine needs a HIP device (or an injected `kernels` backend for CPU rehearsal)")
            kernels = HipKernels(self.device.index or 0)
        self.k = kernels
        self._z = lambda *s: torch.zeros(*s, dtype=self.t_dtype, device=self.device)
        z = self._z
        # every block is padded to n_max columns so that all ring messages have one size
        self.pm = z(4, self.n_max)     # x y z m of the owned sinks == this rank's source chunk
        self.vel = z(3, self.n_max)
        self.acc = z(3, self.n_max)
        self.ring = [z(4, self.n_max), z(4, self.n_max)] if self.world > 1 else []
        self.gathered = z(self.world, 4, self.n_max) if (self.world > 1 and mode == "allgather") else None
        self.out, self.back, self.out_phi, self.back_phi = [], [], [], []
        self.scal = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.phi = None                # per-sink potential of the fused force+potential kernels (step_with_energy)
        self._acc_key = None
        self.uniform_mass = False
        self.own_chunk_symmetric = False   # the last accel() ran the pair-symmetric kernel on this rank's own chunk (HIP backend only)
        self.kernel_timer = None   # optional object with begin()/end(interactions): brackets every accel launch
        self.hop_probe = None      # optional HopProbe: per-hop exchange time next to the kernel it overlaps
        self.heartbeat = None      # optional callable(label): called as the schedule advances (bench.py's stall watchdog)
        if pairwise is None:
            import os
            pairwise = self.world > 1 and hasattr(kernels, "accel_pair") and os.environ.get("NB_PAIRWISE_SCHEDULE", "1") != "0"
        self.set_schedule(self._agree(bool(pairwise)))

    # -- schedule -------------------------------------------------------------------------------
    def _agree(self, flag: bool) -> bool:
        """The ranks must post the same message sequence: take the MIN of the flag over the group (a rank that resolved the default from a
        different environment would otherwise deadlock the others in batch_isend_irecv instead of failing with an error)."""
        dist = self._dist
        if not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return flag
        t = self._torch.tensor([1 if flag else 0], dtype=self._torch.int32,
                               device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def set_schedule(self, pairwise: bool) -> None:
        """Switch a live engine between the pair-symmetric and the one-sided schedule (same call on every rank). State is kept; the cached
        accelerations stay valid (both schedules produce them). Buffers of the return leg are allocated the first time they are needed."""
        self.pairwise = bool(pairwise) and self.world > 1
        self.H = (self.world - 1) // 2 if self.pairwise else 0
        # an even ring's chunk half-way round belongs to two ranks at once: each takes half of ITS pairs two-sidedly (the lower rank its own
        # lower half against the whole visiting chunk, the higher rank its whole chunk against the visitor's upper half) and they swap partials
        self.split_half = self.pairwise and self.world % 2 == 0
        nbuf = self.H + (1 if self.split_half else 0)
        # partial sums of the visiting chunk (sent back to its owner; two buffers in flight) and the partials of the own chunk returned by the
        # ranks that were handed it
        if nbuf > 0 and not self.out:
            self.out = [self._z(3, self.n_max), self._z(3, self.n_max)]
        while len(self.back) < nbuf:
            self.back.append(self._z(3, self.n_max))

    # -- state ----------------------------------------------------------------------------------
    def set_state(self, state: State) -> None:
        """`state` is the GLOBAL state (same on every rank); each rank keeps its own slice."""
        if state.n != self.n_total:
            raise ValueError(f"state has {state.n} particles, engine was built for {self.n_total}")
        t = self._torch
        s = slice(self.off, self.off + self.n_loc)
        pm = np.zeros((4, self.n_max), self.np_dtype)
        pm[:3, : self.n_loc] = np.asarray(state.pos)[s].T
        pm[3, : self.n_loc] = np.asarray(state.mass)[s]
        vel = np.zeros((3, self.n_max), self.np_dtype)
        vel[:, : self.n_loc] = np.asarray(state.vel)[s].T
        # equal masses everywhere (checked on the global host copy, identical on every rank): cheaper inner loop
        m_all = np.asarray(state.mass)
        self.uniform_mass = bool(m_all.size > 0 and np.all(m_all == m_all.flat[0]))
        self.pm.copy_(t.from_numpy(pm))
        self.vel.copy_(t.from_numpy(vel))
        self._acc_key = None

    def gather_state(self) -> State:
        """Global state on every rank (all-gather of the padded blocks, padding stripped)."""
        t, dist = self._torch, self._dist
        blocks = []
        for src in (self.pm, self.vel, self.acc):
            if self.world > 1:
                out = t.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
                dist.all_gather_into_tensor(out.view(-1), src.contiguous().view(-1), group=self.group)
            else:
                out = src.unsqueeze(0)
            blocks.append(out.cpu().numpy())
        pm, vel, acc = blocks
        cat = lambda b, rows: np.concatenate([b[r, rows, : self.counts[r]] for r in range(self.world)], axis=1).T
        st = State(np.ascontiguousarray(cat(pm, slice(0, 3))), np.ascontiguousarray(cat(vel, slice(0, 3))),
                   np.concatenate([pm[r, 3, : self.counts[r]] for r in range(self.world)]))
        st.acc = np.ascontiguousarray(cat(acc, slice(0, 3))) if self._acc_key is not None else None
        return st

    # -- transport of one hop (overridden by tools/virtual_ring.py) ---------------------------------
    def _glob(self, k: int) -> int:
        return self._dist.get_global_rank(self.group, k) if self.group is not None else k

    def _unordered(self) -> bool:
        """RCCL/NCCL collectives are ordered after the work already enqueued on the compute stream, so a receive into a buffer the
        previous kernel is still reading is safe. Other backends (gloo moving device tensors: the one-GPU rehearsal of
        tests/test_gpu_dist.py) give no such ordering: the compute stream is drained before every exchange there."""
        return self.device.type == "cuda" and self._dist.get_backend(self.group) != "nccl"

    def _fence(self) -> None:
        if self._unordered():
            self._torch.cuda.current_stream(self.device).synchronize()

    def _post_hop(self, h: int, cur):
        """Position hop h: the block in hand goes on to rank + 1, the block of rank r - h - 1 arrives from rank - 1. Returns (requests,
        receive buffer); posted BEFORE the kernel that works on `cur` so that it runs beside it."""
        dist = self._dist
        P, r = self.world, self.rank
        recv_buf = self.ring[h & 1]
        ops = [dist.P2POp(dist.isend, cur, self._glob((r + 1) % P), self.group),
               dist.P2POp(dist.irecv, recv_buf, self._glob((r - 1) % P), self.group)]
        return dist.batch_isend_irecv(ops), recv_buf

    def _post_return(self, h: int, eps: float, G: float, with_potential: bool):
        """Return leg of two-sided hop h: the visitor's partial sums (self.out[h & 1]) go to their owner, rank r - h; this rank's partial
        comes from the rank that holds ITS chunk at this hop, rank r + h, into self.back[h - 1]."""
        dist = self._dist
        P, r = self.world, self.rank
        to, frm = self._glob((r - h) % P), self._glob((r + h) % P)
        ops = [dist.P2POp(dist.isend, self.out[h & 1], to, self.group), dist.P2POp(dist.irecv, self.back[h - 1], frm, self.group)]
        if with_potential:
            ops += [dist.P2POp(dist.isend, self.out_phi[h & 1], to, self.group), dist.P2POp(dist.irecv, self.back_phi[h - 1], frm, self.group)]
        return dist.batch_isend_irecv(ops)

    def _beat(self, label: str) -> None:
        if self.heartbeat is not None:
            self.heartbeat(label)

    # -- force evaluation -----------------------------------------------------------------------
    def _for_each_chunk(self, fn) -> None:
        """Call fn(src_block, owner_rank, first) for the own chunk and then every other rank's chunk."""
        dist = self._dist
        P, r = self.world, self.rank
        if P == 1:
            fn(self.pm, r, True)
            return
        host_blocking = self._unordered() or self.device.type != "cuda"
        if self.mode == "allgather":
            self._fence()
            work = dist.all_gather_into_tensor(self.gathered.view(-1), self.pm.view(-1), group=self.group, async_op=True)
            fn(self.pm, r, True)          # own chunk overlaps the collective
            work.wait()
            for h in range(1, P):
                o = (r - h) % P
                fn(self.gathered[o], o, False)
            return
        cur = self.pm
        for h in range(P):
            reqs = None
            if h < P - 1:
                self._fence()
                if self.hop_probe is not None and self.hop_probe.enabled:
                    self.hop_probe.posting()
                reqs, recv_buf = self._post_hop(h, cur)
            probe = self.hop_probe if (reqs is not None and self.hop_probe is not None and self.hop_probe.enabled) else None
            if probe is not None:
                probe.exchange(reqs, lambda: fn(cur, (r - h) % P, h == 0), host_blocking=host_blocking)
                cur = recv_buf
                continue
            fn(cur, (r - h) % P, h == 0)
            if reqs is not None:
                for q in reqs:
                    q.wait()
                cur = recv_buf

    def _accel_pairwise(self, eps: float, G: float, with_potential: bool) -> None:
        """The pair-symmetric schedule (see __init__): own chunk (symmetric inside the kernel), then the chunks of ranks r-1 .. r-H
        two-sidedly with the partial sums sent back to their owners, then — P even — the chunk half-way round, whose pairs this rank shares
        with that chunk's owner (half each, two-sidedly)."""
        t, dist = self._torch, self._dist
        P, r, H = self.world, self.rank, self.H
        half = P % 2 == 0
        hops = H + (1 if half else 0)                  # chunks that have to arrive
        tm = self.kernel_timer
        host_blocking = self._unordered() or self.device.type != "cuda"
        if with_potential and len(self.back_phi) < len(self.back):
            zf = lambda: t.zeros(self.n_max, dtype=t.float64, device=self.device)
            self.out_phi = self.out_phi or [zf(), zf()]
            while len(self.back_phi) < len(self.back):
                self.back_phi.append(zf())

        gather = None
        if self.mode == "allgather":
            self._fence()
            gather = dist.all_gather_into_tensor(self.gathered.view(-1), self.pm.view(-1), group=self.group, async_op=True)

        def kernel(h, cur):
            owner = (r - h) % P
            if tm is not None:
                tm.begin()
            if h > H and self.split_half:                # the chunk half-way round: half of its pairs, two-sidedly (see __init__)
                n_o = self.counts[owner]
                if r < owner:                            # lower rank: own lower half x the whole visiting chunk
                    n_lo = self.n_loc // 2
                    self.k.accel_pair(self.pm[:, :n_lo], n_lo, cur, n_o, eps, G, self.acc[:, :n_lo], self.out[h & 1], True, self.uniform_mass,
                                      self.phi[:n_lo] if with_potential else None, self.out_phi[h & 1] if with_potential else None)
                    delivered = 2.0 * n_lo * n_o
                else:                                    # higher rank: whole own chunk x the visitor's upper half
                    lo_o = n_o // 2
                    self.k.accel_pair(self.pm, self.n_loc, cur[:, lo_o:n_o], n_o - lo_o, eps, G, self.acc, self.out[h & 1], True, self.uniform_mass,
                                      self.phi if with_potential else None, self.out_phi[h & 1] if with_potential else None)
                    delivered = 2.0 * self.n_loc * (n_o - lo_o)
                if tm is not None:
                    tm.end(delivered)
                return
            if h == 0:                                   # own chunk: this rank's sinks only (pair-symmetric inside the kernel)
                if with_potential:
                    self.k.accel_pot(self.pm, self.n_loc, cur, self.counts[owner], eps, G, self.acc, self.phi, False, self.uniform_mass)
                else:
                    self.k.accel(self.pm, self.n_loc, cur, self.counts[owner], eps, G, self.acc, False, self.uniform_mass)
                if hasattr(self.k, "last_accel_symmetric"):
                    self.own_chunk_symmetric = self.k.last_accel_symmetric()
            else:                                        # every pair of (own chunk, visiting chunk) once; the visitor's sums go home
                self.k.accel_pair(self.pm, self.n_loc, cur, self.counts[owner], eps, G, self.acc, self.out[h & 1], True, self.uniform_mass,
                                  self.phi if with_potential else None, self.out_phi[h & 1] if with_potential else None)
            if tm is not None:                           # interactions DELIVERED by the launch: both directions of a two-sided hop
                tm.end(float(self.n_loc) * float(self.counts[owner]) * (2.0 if 0 < h <= H else 1.0))

        hp = self.hop_probe if (self.hop_probe is not None and self.hop_probe.enabled) else None

        def wait_back(reqs):
            if hp is not None:
                hp.wait_back(reqs)
            else:
                for q in reqs:
                    q.wait()

        cur = self.pm
        back_reqs = [None] * (hops + 1)
        two_sided = H + (1 if self.split_half else 0)    # hops 1 .. two_sided return a partial
        for h in range(hops + 1):
            reqs = None
            if h < hops and self.mode == "ring":
                self._fence()
                if hp is not None:
                    hp.posting()
                reqs, recv_buf = self._post_hop(h, cur)
            if 2 <= h <= two_sided and back_reqs[h - 2] is not None:  # the partial buffer of hop h - 2 is about to be overwritten: its send must be through
                wait_back(back_reqs[h - 2])
                back_reqs[h - 2] = None
            probe = hp if reqs is not None else None
            if probe is not None:
                probe.exchange(reqs, lambda: kernel(h, cur), host_blocking=host_blocking)
            else:
                kernel(h, cur)
            if 1 <= h <= two_sided:                      # the visitor's partial goes to its owner; mine comes from the rank holding my chunk now
                self._fence()
                mark = hp._mark() if hp is not None else None
                back_reqs[h] = self._post_return(h, eps, G, with_potential)
                if hp is not None:
                    hp.returning(mark, back_reqs[h], host_blocking)
            if probe is None and reqs is not None:
                for q in reqs:
                    q.wait()
            if h < hops:
                if self.mode == "ring":
                    cur = recv_buf
                else:
                    if gather is not None:
                        gather.wait()
                        gather = None
                    cur = self.gathered[(r - h - 1) % P]
            self._beat("hop %d" % h)
        for reqs in back_reqs:
            if reqs is not None:
                wait_back(reqs)

        def closing():
            for h in range(1, H + 1):                    # fixed order: bit-reproducible
                self.acc[:, : self.n_loc] += self.back[h - 1][:, : self.n_loc]
                if with_potential:
                    self.phi[: self.n_loc] += self.back_phi[h - 1][: self.n_loc]
            if self.split_half:                          # what the rank half-way round worked out for this one
                lo = self.n_loc // 2 if r < (r + P // 2) % P else 0        # the lower rank did its own lower half itself
                k = self.n_loc - lo
                self.acc[:, lo: self.n_loc] += self.back[H][:, :k]
                if with_potential:
                    self.phi[lo: self.n_loc] += self.back_phi[H][:k]
        if hp is not None:
            hp.closing(closing)
        else:
            closing()

    def _tail_fused(self) -> bool:
        """One rank owning every particle: the force launch can apply the leapfrog tail itself (HipKernels.accel_kdk)."""
        return self.world == 1 and not self.pairwise and hasattr(self.k, "accel_kdk")

    def accel(self, eps: float, G: float = 1.0, with_potential: bool = False, tail=None) -> None:
        """acc <- a(x) of the owned sinks due to ALL particles; with_potential also accumulates self.phi over the chunks. `tail` =
        (kick_close, kick_open, drift), one rank only: the leapfrog tail applied with the force (step())."""
        tm = self.kernel_timer
        assert tail is None or (self._tail_fused() and not with_potential)
        if with_potential and self.phi is None:
            self.phi = self._torch.zeros(self.n_max, dtype=self._torch.float64, device=self.device)
        if self.pairwise:
            self._accel_pairwise(eps, G, with_potential)
            self._acc_key = (eps, G)
            return

        def one(src, owner, first):
            if tm is not None:
                tm.begin()
            if with_potential:
                self.k.accel_pot(self.pm, self.n_loc, src, self.counts[owner], eps, G, self.acc, self.phi, not first, self.uniform_mass)
            elif tail is not None:
                self.k.accel_kdk(self.pm, self.n_loc, src, self.counts[owner], eps, G, self.acc, not first, self.vel, *tail, uniform_mass=self.uniform_mass)
            else:
                self.k.accel(self.pm, self.n_loc, src, self.counts[owner], eps, G, self.acc, not first, self.uniform_mass)
            if tm is not None:
                tm.end(float(self.n_loc) * float(self.counts[owner]))
            if first and hasattr(self.k, "last_accel_symmetric"):
                self.own_chunk_symmetric = self.k.last_accel_symmetric()
        self._for_each_chunk(one)
        self._acc_key = (eps, G)

    def step(self, dt: float, nsteps: int = 1, eps: float = 0.0, G: float = 1.0) -> None:
        if nsteps <= 0:
            return
        if self._acc_key != (eps, G):
            self.accel(eps, G)
        h = 0.5 * dt
        if self._tail_fused():
            # one rank: the kernel that completes the accelerations also applies the closing half kick and, between the steps of one call,
            # the next opening half kick and drift — two launches per step after the first (include/nbody_amd.h: nb_accel_kdk_dev)
            self.k.kick_drift(self.pm, self.vel, self.acc, self.n_loc, h, dt)
            for s in range(int(nsteps)):
                self.accel(eps, G, tail=(h, 0.0, 0.0) if s == nsteps - 1 else (h, h, dt))
            return
        kd = getattr(self.k, "kick_drift", None)
        for _ in range(int(nsteps)):
            if kd is not None:
                kd(self.pm, self.vel, self.acc, self.n_loc, h, dt)
            else:
                self.k.kick(self.vel, self.acc, self.n_loc, h)
                self.k.drift(self.pm, self.vel, self.n_loc, dt)
            self.accel(eps, G)
            self.k.kick(self.vel, self.acc, self.n_loc, h)

    def step_with_energy(self, dt: float, nsteps: int = 1, eps: float = 0.0, G: float = 1.0):
        """KDK steps returning (kinetic[nsteps], potential[nsteps]) of the WHOLE system after every step. The potential is a
        by-product of the force kernels (`nb_accel_pot_dev` accumulates it over the source chunks as they arrive), so the energy
        log costs one all-reduce of 2 x nsteps doubles at the end instead of a second ring pass per diagnostic."""
        t = self._torch
        log = t.zeros((max(int(nsteps), 0), 2), dtype=t.float64, device=self.device)
        if nsteps <= 0:
            return np.empty(0), np.empty(0)
        if self._acc_key != (eps, G):
            self.accel(eps, G)
        h = 0.5 * dt
        for s in range(int(nsteps)):
            self.k.kick(self.vel, self.acc, self.n_loc, h)
            self.k.drift(self.pm, self.vel, self.n_loc, dt)
            self.accel(eps, G, with_potential=True)
            self.k.kick(self.vel, self.acc, self.n_loc, h)
            self.k.kinetic(self.vel, self.pm[3], self.n_loc, log[s, 0:1])
            self.k.pot_energy(self.phi, self.pm[3], self.n_loc, eps, log[s, 1:2], self.uniform_mass)
        if self.world > 1:
            self._dist.all_reduce(log, group=self.group)
        out = log.cpu().numpy()
        return out[:, 0].copy(), -0.5 * G * out[:, 1]

    def energy(self, eps: float, G: float = 1.0) -> Tuple[float, float]:
        self.k.kinetic(self.vel, self.pm[3], self.n_loc, self.scal[0:1])

        def one(src, owner, first):
            self.k.potential(self.pm, self.n_loc, self.off, src, self.counts[owner], self.offsets[owner], eps,
                             self.scal[1:2], not first)
        self._for_each_chunk(one)
        if self.world > 1:
            self._dist.all_reduce(self.scal, group=self.group)
        k, w = self.scal.cpu().tolist()
        return k, -0.5 * G * w

    def local_acc(self) -> np.ndarray:
        return self.acc[:, : self.n_loc].T.contiguous().cpu().numpy()
