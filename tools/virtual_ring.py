"""Rehearsal scaffolding (NOT product code): a P-rank ring played by ONE real RCCL rank.

RCCL refuses two ranks on one device, but a single rank may send to and receive from itself. `VirtualRingEngine` is a
`ShardedEngine` whose two transport hooks are replaced: the one real rank plays rank 0 of a P-rank ring — it owns the first
of P sink ranges, holds the other (virtual) ranks' source blocks locally, and at every hop sends the block its predecessor
would have sent TO ITSELF through the communicator. Everything the RCCL path of the production schedule relies on is then
exercised for real — batch_isend_irecv posted before the kernel, the communicator stream's ordering against the compute
stream, wait(), receive-buffer reuse, the return leg of the pair-symmetric schedule INCLUDING the split half-way hop of an
even ring (what the other owner would return is computed here and sent through the communicator; this rank's own partial
for the visitor is sent too, into a sink buffer, so the `out` buffers are re-used under the communicator's ordering exactly
as in a real run) — except the xGMI link. Only accel() / local_acc() make sense in this mode.

Used by tests/test_gpu_rccl.py and tools/rccl_virtual_ring.py."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bachelor_thesis_n_body_amd.dist import ShardedEngine  # noqa: E402


class VirtualRingEngine(ShardedEngine):
    def __init__(self, n_total, dtype=np.float32, device=None, virtual_world=2, pairwise=False, kernels=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size() != 1 or virtual_world < 2:
            raise ValueError("VirtualRingEngine needs an initialised single-rank process group and virtual_world >= 2")
        super().__init__(n_total, dtype, device=device, kernels=kernels, mode="ring", pairwise=pairwise, world=int(virtual_world))
        z = self._z
        self.virtual = z(self.world, 4, self.n_max)      # the source blocks of the other (virtual) ranks
        # what the virtual owners would return, pre-allocated: one payload per returned partial in flight (two), a scratch for the A side
        # of the stand-in kernel, and the sink the real rank's own outgoing partial lands in
        self._payload = [z(3, self.n_max), z(3, self.n_max)]
        self._scratch = z(3, self.n_max)
        self._sink = [z(3, self.n_max), z(3, self.n_max)]
        f64 = lambda: torch.zeros(self.n_max, dtype=torch.float64, device=self.device)
        self._payload_phi, self._scratch_phi, self._sink_phi = [f64(), f64()], f64(), [f64(), f64()]
        self._self = dist.get_rank()

    def set_state(self, state):
        super().set_state(state)
        allpm = np.zeros((self.world, 4, self.n_max), self.np_dtype)
        for o in range(self.world):
            so = slice(self.offsets[o], self.offsets[o] + self.counts[o])
            allpm[o, :3, : self.counts[o]] = np.asarray(state.pos)[so].T
            allpm[o, 3, : self.counts[o]] = np.asarray(state.mass)[so]
        self.virtual.copy_(self._torch.from_numpy(allpm))

    def _post_hop(self, h, cur):
        dist = self._dist
        recv_buf = self.ring[h & 1]
        send_buf = self.virtual[(self.rank - h - 1) % self.world]     # what the predecessor would have sent at this hop
        ops = [dist.P2POp(dist.isend, send_buf, self._self, self.group), dist.P2POp(dist.irecv, recv_buf, self._self, self.group)]
        return dist.batch_isend_irecv(ops), recv_buf

    def _post_return(self, h, eps, G, with_potential):
        dist = self._dist
        P, r, H = self.world, self.rank, self.H
        o = (r + h) % P                                   # the rank holding this rank's chunk at hop h: its partial is computed here
        payload, pphi = self._payload[h & 1], (self._payload_phi[h & 1] if with_potential else None)
        sphi = self._scratch_phi if with_potential else None
        if h <= H:                                        # a full two-sided hop: o's whole chunk x this rank's whole chunk
            self.k.accel_pair(self.virtual[o], self.counts[o], self.pm, self.n_loc, eps, G, self._scratch, payload, False, self.uniform_mass, sphi, pphi)
        else:                                             # the half-way hop of an even ring; rank 0 is the LOWER rank of the pair, so the other
            lo = self.n_loc // 2                          # owner takes its whole chunk x this rank's upper half and returns that half's sums
            k = self.n_loc - lo
            self.k.accel_pair(self.virtual[o], self.counts[o], self.pm[:, lo: self.n_loc], k, eps, G, self._scratch, payload, False,
                              self.uniform_mass, sphi, pphi)
        # two send/receive pairs with itself, matched in order: this rank's outgoing partial (dropped into a sink) and the stand-in's payload
        ops = [dist.P2POp(dist.isend, self.out[h & 1], self._self, self.group), dist.P2POp(dist.irecv, self._sink[h & 1], self._self, self.group),
               dist.P2POp(dist.isend, payload, self._self, self.group), dist.P2POp(dist.irecv, self.back[h - 1], self._self, self.group)]
        if with_potential:
            ops += [dist.P2POp(dist.isend, self.out_phi[h & 1], self._self, self.group), dist.P2POp(dist.irecv, self._sink_phi[h & 1], self._self, self.group),
                    dist.P2POp(dist.isend, pphi, self._self, self.group), dist.P2POp(dist.irecv, self.back_phi[h - 1], self._self, self.group)]
        return dist.batch_isend_irecv(ops)
