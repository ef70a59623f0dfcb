// Multi-device context of the C-ABI ("nb_create(..., ndev, dev_ids)", §8e sink decomposition) — nb_create_multi.
//
// One host thread drives ndev GPUs of one node. Device g owns the contiguous sink range [off_g, off_g + cnt_g) as an ordinary
// single-device context (its own stream, SoA state, scratch) and integrates only those particles. Every device needs ALL
// sources: at each force evaluation it PULLS the (x,y,z,m) block of every other owner straight out of that owner's state with
// hipMemcpyPeerAsync on a copy stream — one contiguous 16*cnt (fp32) / 32*cnt (fp64) byte message per owner — while its force
// kernel works on the block that arrived before (two receive buffers). At hop h device g reads owner (g + h) mod P, so in every hop
// each device is read by exactly one other: on the fully connected xGMI mesh of an MI355X node the P pulls of a hop run on P
// different links (a ring that forwards blocks would use one link per device). Accelerations need no reduction — each sink's
// sum completes locally, chunk by chunk (NB_ACCUMULATE); energies are summed on the host from P device scalars.
//
// Ordering, all by events (no host synchronisation inside a step):
//   ev_pos[o]      recorded on owner o's stream when its positions are final          -> a pull of o's block waits for it
//   ev_copy[g][b]  recorded on g's copy stream after the pull into receive buffer b    -> the kernel reading b waits for it
//   ev_kern[g][b]  recorded on g's stream after the kernel that read buffer b          -> the next pull into b waits for it
//   ev_pulled[o][g] recorded on g's copy stream after pulling o's block                -> o's next drift waits for all of them
// The same device may appear several times in dev_ids (a one-GPU rehearsal of the P-device schedule: tests/test_gpu_multi.py).
//
// Pair-symmetric schedule (default; NB_OPT_PAIR_SYMMETRY 0 restores the one above). Every pair of chunks is worked on by ONE of its two
// owners: device g pulls only the blocks of owners g+1 .. g+(P-1)/2 and evaluates every (own, visiting) pair ONCE (nb_launch_accel_pair:
// the two-block form of the pair-symmetric kernels), adding to its own sums and collecting the visitor's partial sums in a buffer that it
// then PUSHES to the visitor's owner (hipMemcpyPeerAsync on its compute stream, behind the kernel); for an even P the block half-way round
// is met from both sides one-sidedly; each owner finally adds the (P-1)/2 partials it was sent, in a fixed order. P/2 pulled blocks per
// device instead of P-1, half the pair work. Two more events:
//   ev_back[o][h]  recorded on the pusher's stream after the partial for owner o from hop h has landed   -> o's add waits for it
//   ev_added[o]    recorded on o's stream after it has added its partials                                 -> the next push into o's buffers waits for it
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <vector>

struct nb_multi_state {
  int P = 0;
  std::vector<int> dev;
  std::vector<nb_ctx*> shard;
  std::vector<int64_t> cnt, off;
  int64_t n_max = 0;
  std::vector<void*> recv0, recv1;                    // two receive buffers of 4 * n_max elements per device
  std::vector<hipStream_t> copy;                      // per-device copy stream
  std::vector<hipEvent_t> ev_pos;                     // [P]
  std::vector<hipEvent_t> ev_copy, ev_kern;           // [P][2]
  std::vector<hipEvent_t> ev_pulled;                  // [owner][reader]
  std::vector<char> kern_valid;                       // ev_kern[g][b] has been recorded (also by an EARLIER force evaluation)
  std::vector<char> pulled_valid;                     // ev_pulled[o][g] has been recorded since o's positions last changed
  // pair-symmetric schedule: H = (P - 1) / 2 returned partials per owner
  int H = 0, R = 0;                                   // R = H + 1 for an even P: the chunk half-way round is split between its two owners
  std::vector<void*> outb;                            // [P]   3 * n_max elements: the visiting chunk's partial accelerations
  std::vector<double*> outphi;                        // [P]   n_max doubles (lazy: energy runs only)
  std::vector<void*> back;                            // [P*R] 3 * n_max elements: partials of the own chunk returned by the devices that were handed it
  std::vector<double*> backphi;                       // [P*R]
  std::vector<hipEvent_t> ev_back;                    // [P*R]
  std::vector<hipEvent_t> ev_added;                   // [P]
  std::vector<char> added_valid;
  long long pairwise = -1;                            // NB_OPT_PAIR_SYMMETRY of the context (-1: default = on)
  bool own_chunk_sym = false;                         // device 0's own-chunk launch of the last force evaluation ran the pair-symmetric kernel
  bool uniform = false;
  bool acc_valid = false;
  double acc_eps = -1.0, acc_G = 0.0;
};

namespace {

inline size_t esz(int dtype) { return dtype == NB_F64 ? 8 : 4; }

#define MHIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return nb_fail((c), NB_ERR_HIP, std::string("multi: ") + #call + ": " + hipGetErrorString(e_)); } while (0)

inline char* field_ptr(nb_ctx* s, int field) { return reinterpret_cast<char*>(s->state) + (size_t)field * (size_t)s->n * esz(s->dtype); }

// fn(g, owner, src_base, first): src_base = device pointer on device g to the owner's [x | y | z | m] block (cnt[owner] elements each)
template <typename F>
int for_each_chunk(nb_ctx* c, F fn) {
  nb_multi_state& m = *c->multi;
  const int P = m.P;
  const size_t es = esz(c->dtype);
  for (int g = 0; g < P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    MHIP(c, hipEventRecord(m.ev_pos[g], m.shard[g]->stream));
  }
  for (int h = 0; h < P; ++h) {
    for (int g = 0; g < P; ++g) {
      MHIP(c, hipSetDevice(m.dev[g]));
      nb_ctx* s = m.shard[g];
      if (h + 1 < P) {                                   // pull the block of hop h+1 while hop h computes
        const int o2 = (g + h + 1) % P, b = (h + 1) & 1;
        void* dst = b ? m.recv1[g] : m.recv0[g];
        MHIP(c, hipStreamWaitEvent(m.copy[g], m.ev_pos[o2], 0));
        // the last kernel that read this buffer (hop h-1, or the tail of the PREVIOUS evaluation) must be done with it
        if (m.kern_valid[g * 2 + b]) MHIP(c, hipStreamWaitEvent(m.copy[g], m.ev_kern[g * 2 + b], 0));
        MHIP(c, hipMemcpyPeerAsync(dst, m.dev[g], m.shard[o2]->state, m.dev[o2], 4 * (size_t)m.cnt[o2] * es, m.copy[g]));
        MHIP(c, hipEventRecord(m.ev_copy[g * 2 + b], m.copy[g]));
        MHIP(c, hipEventRecord(m.ev_pulled[o2 * P + g], m.copy[g]));
        m.pulled_valid[o2 * P + g] = 1;
      }
      const int o = (g + h) % P, b = h & 1;
      const void* src = h == 0 ? s->state : (b ? m.recv1[g] : m.recv0[g]);
      if (h >= 1) MHIP(c, hipStreamWaitEvent(s->stream, m.ev_copy[g * 2 + b], 0));
      int rc = fn(g, o, reinterpret_cast<const char*>(src), h == 0);
      if (rc) return nb_fail(c, rc, std::string("multi: device ") + std::to_string(m.dev[g]) + ": " + s->err);
      if (h >= 1) { MHIP(c, hipEventRecord(m.ev_kern[g * 2 + b], s->stream)); m.kern_valid[g * 2 + b] = 1; }
    }
  }
  return NB_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) add_flat_kernel(T* __restrict__ a, const T* __restrict__ b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] += b[i];
}

// The pair-symmetric schedule (file header): own chunk, then the chunks of owners g+1 .. g+H two-sidedly with the visitor's partial sums
// pushed to its owner, then — P even — the chunk half-way round one-sidedly; finally every owner adds the partials it was sent.
int refresh_acc_pairwise(nb_ctx* c, double eps, double G, bool with_phi) {
  nb_multi_state& m = *c->multi;
  const int P = m.P, H = m.H, R = m.R, hops = R;        // R = H (+ 1: the chunk half-way round an even ring, split between its two owners)
  const size_t es = esz(c->dtype);
  for (int g = 0; g < P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    MHIP(c, hipEventRecord(m.ev_pos[g], m.shard[g]->stream));
    if (with_phi && R > 0 && !m.outphi[g]) {
      MHIP(c, hipMalloc((void**)&m.outphi[g], (size_t)m.n_max * sizeof(double)));
      for (int h = 0; h < R; ++h) MHIP(c, hipMalloc((void**)&m.backphi[g * R + h], (size_t)m.n_max * sizeof(double)));
    }
  }
  for (int h = 0; h <= hops; ++h) {
    for (int g = 0; g < P; ++g) {
      MHIP(c, hipSetDevice(m.dev[g]));
      nb_ctx* s = m.shard[g];
      if (h < hops) {                                    // pull the block of hop h+1 while hop h computes
        const int o2 = (g + h + 1) % P, b = (h + 1) & 1;
        void* dst = b ? m.recv1[g] : m.recv0[g];
        MHIP(c, hipStreamWaitEvent(m.copy[g], m.ev_pos[o2], 0));
        if (m.kern_valid[g * 2 + b]) MHIP(c, hipStreamWaitEvent(m.copy[g], m.ev_kern[g * 2 + b], 0));
        MHIP(c, hipMemcpyPeerAsync(dst, m.dev[g], m.shard[o2]->state, m.dev[o2], 4 * (size_t)m.cnt[o2] * es, m.copy[g]));
        MHIP(c, hipEventRecord(m.ev_copy[g * 2 + b], m.copy[g]));
        MHIP(c, hipEventRecord(m.ev_pulled[o2 * P + g], m.copy[g]));
        m.pulled_valid[o2 * P + g] = 1;
      }
      const int o = (g + h) % P, b = h & 1;
      const char* src = reinterpret_cast<const char*>(h == 0 ? s->state : (b ? m.recv1[g] : m.recv0[g]));
      if (h >= 1) MHIP(c, hipStreamWaitEvent(s->stream, m.ev_copy[g * 2 + b], 0));
      const size_t so = (size_t)m.cnt[o] * es;
      const int uni = m.uniform ? NB_UNIFORM_MASS : 0;
      double* phi = with_phi ? s->phi : nullptr;
      int rc;
      if (h == 0) {                                      // own chunk (symmetric inside the launcher)
        const int flags = (h == 0 ? 0 : NB_ACCUMULATE) | uni;
        if (c->dtype == NB_F32)
          rc = nb_launch_accel_f32(s, (const float*)field_ptr(s, 0), (const float*)field_ptr(s, 1), (const float*)field_ptr(s, 2), s->n,
                                   (const float*)src, (const float*)(src + so), (const float*)(src + 2 * so), (const float*)(src + 3 * so), m.cnt[o],
                                   (float)(eps * eps), (float)G, (float*)field_ptr(s, 7), (float*)field_ptr(s, 8), (float*)field_ptr(s, 9), flags, phi);
        else
          rc = nb_launch_accel_f64(s, (const double*)field_ptr(s, 0), (const double*)field_ptr(s, 1), (const double*)field_ptr(s, 2), s->n,
                                   (const double*)src, (const double*)(src + so), (const double*)(src + 2 * so), (const double*)(src + 3 * so), m.cnt[o],
                                   eps * eps, G, (double*)field_ptr(s, 7), (double*)field_ptr(s, 8), (double*)field_ptr(s, 9), flags, phi);
        if (rc) return nb_fail(c, rc, std::string("multi: device ") + std::to_string(m.dev[g]) + ": " + s->err);
        if (g == 0) m.own_chunk_sym = s->last_accel_sym;
      } else {                                           // every (own, visiting) pair once; the visitor's sums go to its owner
        // hop <= H: the whole own chunk x the whole visiting chunk. The hop half-way round an even ring (h = H + 1) is shared with the
        // visitor's owner: the lower device takes its own LOWER half x the whole visitor, the higher device its whole chunk x the visitor's
        // UPPER half — together every pair of the two chunks once, half each.
        int64_t a0 = 0, na = s->n, b0 = 0, nb = m.cnt[o];
        if (h > H) { if (g < o) na = s->n / 2; else { b0 = m.cnt[o] / 2; nb = m.cnt[o] - b0; } }
        const size_t ao = (size_t)a0 * es, bo = (size_t)b0 * es, sb = (size_t)nb * es;
        char* ob = reinterpret_cast<char*>(m.outb[g]);
        rc = nb_launch_accel_pair(s, c->dtype, field_ptr(s, 0) + ao, field_ptr(s, 1) + ao, field_ptr(s, 2) + ao, field_ptr(s, 3) + ao, na,
                                  src + bo, src + so + bo, src + 2 * so + bo, src + 3 * so + bo, nb, eps * eps, G,
                                  field_ptr(s, 7) + ao, field_ptr(s, 8) + ao, field_ptr(s, 9) + ao, ob, ob + sb, ob + 2 * sb,
                                  phi, with_phi ? m.outphi[g] : nullptr, NB_ACCUMULATE | uni);
        if (rc) return nb_fail(c, rc, std::string("multi: device ") + std::to_string(m.dev[g]) + ": " + s->err);
        // o's buffers of this hop still hold the partial of the PREVIOUS evaluation until o has added it
        if (m.added_valid[o]) MHIP(c, hipStreamWaitEvent(s->stream, m.ev_added[o], 0));
        if (nb > 0) {
          MHIP(c, hipMemcpyPeerAsync(m.back[o * R + h - 1], m.dev[o], ob, m.dev[g], 3 * sb, s->stream));
          if (with_phi) MHIP(c, hipMemcpyPeerAsync(m.backphi[o * R + h - 1], m.dev[o], m.outphi[g], m.dev[g], (size_t)nb * sizeof(double), s->stream));
        }
        MHIP(c, hipEventRecord(m.ev_back[o * R + h - 1], s->stream));
      }
      if (h >= 1) { MHIP(c, hipEventRecord(m.ev_kern[g * 2 + b], s->stream)); m.kern_valid[g * 2 + b] = 1; }
    }
  }
  for (int o = 0; o < P && R > 0; ++o) {                  // every owner adds what it was sent, hop 1 first: a fixed order
    MHIP(c, hipSetDevice(m.dev[o]));
    nb_ctx* s = m.shard[o];
    auto add = [&](void* dst, const void* src, int64_t cnt) {
      if (cnt <= 0) return;
      if (c->dtype == NB_F32) add_flat_kernel<float><<<(int)((cnt + 255) / 256), 256, 0, s->stream>>>((float*)dst, (const float*)src, cnt);
      else add_flat_kernel<double><<<(int)((cnt + 255) / 256), 256, 0, s->stream>>>((double*)dst, (const double*)src, cnt);
    };
    for (int h = 0; h < R; ++h) {
      MHIP(c, hipStreamWaitEvent(s->stream, m.ev_back[o * R + h], 0));
      if (h < H) {                                       // a partial for the whole chunk: [ax | ay | az] is one flat array of 3 n
        add(field_ptr(s, 7), m.back[o * R + h], 3 * s->n);
        if (with_phi && s->n > 0) add_flat_kernel<double><<<(int)((s->n + 255) / 256), 256, 0, s->stream>>>(s->phi, m.backphi[o * R + h], s->n);
      } else {                                           // the half-way hop: the lower device is sent its UPPER half, the higher device its whole chunk
        const int partner = (o + P / 2) % P;
        const int64_t lo = o < partner ? s->n / 2 : 0, k = s->n - lo;
        const char* bsrc = reinterpret_cast<const char*>(m.back[o * R + h]);
        for (int f = 0; f < 3; ++f) add(field_ptr(s, 7 + f) + (size_t)lo * es, bsrc + (size_t)f * (size_t)k * es, k);
        if (with_phi && k > 0) add_flat_kernel<double><<<(int)((k + 255) / 256), 256, 0, s->stream>>>(s->phi + lo, m.backphi[o * R + h], k);
      }
      MHIP(c, hipGetLastError());
    }
    MHIP(c, hipEventRecord(m.ev_added[o], s->stream));
    m.added_valid[o] = 1;
  }
  m.acc_valid = true; m.acc_eps = eps; m.acc_G = G;
  return NB_OK;
}

// owner o is about to change its positions: every pull of its block must have finished
int wait_for_readers(nb_ctx* c, int o) {
  nb_multi_state& m = *c->multi;
  for (int g = 0; g < m.P; ++g)
    if (g != o && m.pulled_valid[o * m.P + g]) {
      MHIP(c, hipStreamWaitEvent(m.shard[o]->stream, m.ev_pulled[o * m.P + g], 0));
      m.pulled_valid[o * m.P + g] = 0;
    }
  return NB_OK;
}

// with_phi: the fused force+potential kernels accumulate shard->phi over the chunks as well (nb_step_energy)
int refresh_acc(nb_ctx* c, double eps, double G, bool with_phi = false) {
  nb_multi_state& m = *c->multi;
  const size_t es = esz(c->dtype);
  if (m.P > 1 && m.pairwise != 0) return refresh_acc_pairwise(c, eps, G, with_phi);
  int rc = for_each_chunk(c, [&](int g, int o, const char* src, bool first) -> int {
    nb_ctx* s = m.shard[g];
    const size_t so = (size_t)m.cnt[o] * es;
    const int flags = (first ? 0 : NB_ACCUMULATE) | (m.uniform ? NB_UNIFORM_MASS : 0);
    double* phi = with_phi ? s->phi : nullptr;
    int rc1;
    if (c->dtype == NB_F32)
      rc1 = nb_launch_accel_f32(s, (const float*)field_ptr(s, 0), (const float*)field_ptr(s, 1), (const float*)field_ptr(s, 2), s->n,
                                (const float*)src, (const float*)(src + so), (const float*)(src + 2 * so), (const float*)(src + 3 * so), m.cnt[o],
                                (float)(eps * eps), (float)G, (float*)field_ptr(s, 7), (float*)field_ptr(s, 8), (float*)field_ptr(s, 9), flags, phi);
    else
      rc1 = nb_launch_accel_f64(s, (const double*)field_ptr(s, 0), (const double*)field_ptr(s, 1), (const double*)field_ptr(s, 2), s->n,
                                (const double*)src, (const double*)(src + so), (const double*)(src + 2 * so), (const double*)(src + 3 * so), m.cnt[o],
                                eps * eps, G, (double*)field_ptr(s, 7), (double*)field_ptr(s, 8), (double*)field_ptr(s, 9), flags, phi);
    if (first && g == 0) m.own_chunk_sym = s->last_accel_sym;     // the later cross-chunk launches of this schedule are one-sided by construction
    return rc1;
  });
  if (rc) return rc;
  m.acc_valid = true; m.acc_eps = eps; m.acc_G = G;
  return NB_OK;
}

int sync_all(nb_ctx* c) {
  nb_multi_state& m = *c->multi;
  for (int g = 0; g < m.P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    MHIP(c, hipStreamSynchronize(m.copy[g]));
    MHIP(c, hipStreamSynchronize(m.shard[g]->stream));
  }
  return NB_OK;
}

}  // namespace

void nb_multi_release(nb_ctx* c) {
  nb_multi_state* m = c->multi;
  if (!m) return;
  for (int g = 0; g < (int)m->shard.size(); ++g) {
    (void)hipSetDevice(m->dev[g]);
    if (g < (int)m->copy.size() && m->copy[g]) { (void)hipStreamSynchronize(m->copy[g]); }
    if (m->shard[g]) { (void)hipStreamSynchronize(m->shard[g]->stream); }
  }
  for (int g = 0; g < (int)m->shard.size(); ++g) {
    (void)hipSetDevice(m->dev[g]);
    if (g < (int)m->recv0.size() && m->recv0[g]) (void)hipFree(m->recv0[g]);
    if (g < (int)m->recv1.size() && m->recv1[g]) (void)hipFree(m->recv1[g]);
    if (g < (int)m->copy.size() && m->copy[g]) (void)hipStreamDestroy(m->copy[g]);
    if (g < (int)m->ev_pos.size() && m->ev_pos[g]) (void)hipEventDestroy(m->ev_pos[g]);
    if (g < (int)m->outb.size() && m->outb[g]) (void)hipFree(m->outb[g]);
    if (g < (int)m->outphi.size() && m->outphi[g]) (void)hipFree(m->outphi[g]);
    if (g < (int)m->ev_added.size() && m->ev_added[g]) (void)hipEventDestroy(m->ev_added[g]);
    for (int h = 0; h < m->R; ++h) {
      const size_t q = (size_t)g * m->R + h;
      if (q < m->back.size() && m->back[q]) (void)hipFree(m->back[q]);
      if (q < m->backphi.size() && m->backphi[q]) (void)hipFree(m->backphi[q]);
      if (q < m->ev_back.size() && m->ev_back[q]) (void)hipEventDestroy(m->ev_back[q]);
    }
    for (int b = 0; b < 2; ++b) {
      if (g * 2 + b < (int)m->ev_copy.size() && m->ev_copy[g * 2 + b]) (void)hipEventDestroy(m->ev_copy[g * 2 + b]);
      if (g * 2 + b < (int)m->ev_kern.size() && m->ev_kern[g * 2 + b]) (void)hipEventDestroy(m->ev_kern[g * 2 + b]);
    }
    for (int o = 0; o < m->P; ++o)
      if (o * m->P + g < (int)m->ev_pulled.size() && m->ev_pulled[o * m->P + g]) (void)hipEventDestroy(m->ev_pulled[o * m->P + g]);
    if (m->shard[g]) (void)nb_destroy(m->shard[g]);
  }
  delete m;
  c->multi = nullptr;
}

int nb_multi_create(nb_ctx* c, int ndev, const int* dev_ids) {
  nb_multi_state* m = new nb_multi_state();
  c->multi = m;
  m->P = ndev;
  m->dev.assign(dev_ids, dev_ids + ndev);
  m->shard.assign(ndev, nullptr);
  m->cnt.resize(ndev); m->off.resize(ndev);
  const int64_t base = c->n / ndev, rem = c->n % ndev;          // the first n % P devices own one more particle (dist.partition)
  for (int g = 0; g < ndev; ++g) { m->cnt[g] = base + (g < rem ? 1 : 0); m->off[g] = g ? m->off[g - 1] + m->cnt[g - 1] : 0; }
  m->n_max = m->cnt[0];
  m->recv0.assign(ndev, nullptr); m->recv1.assign(ndev, nullptr); m->copy.assign(ndev, nullptr);
  m->ev_pos.assign(ndev, nullptr); m->ev_copy.assign(2 * ndev, nullptr); m->ev_kern.assign(2 * ndev, nullptr);
  m->ev_pulled.assign((size_t)ndev * ndev, nullptr); m->pulled_valid.assign((size_t)ndev * ndev, 0); m->kern_valid.assign(2 * (size_t)ndev, 0);
  m->H = (ndev - 1) / 2;
  m->R = ndev > 1 ? ndev / 2 : 0;
  m->outb.assign(ndev, nullptr); m->outphi.assign(ndev, nullptr); m->ev_added.assign(ndev, nullptr); m->added_valid.assign(ndev, 0);
  m->back.assign((size_t)ndev * m->R, nullptr); m->backphi.assign((size_t)ndev * m->R, nullptr); m->ev_back.assign((size_t)ndev * m->R, nullptr);
  const size_t es = esz(c->dtype);
  for (int g = 0; g < ndev; ++g) {
    int rc = nb_create(&m->shard[g], m->cnt[g], c->dtype, dev_ids[g]);
    if (rc) return nb_fail(c, rc, std::string("nb_create_multi: device ") + std::to_string(dev_ids[g]) + ": " + nb_last_error(nullptr));
    MHIP(c, hipSetDevice(dev_ids[g]));
    if (ndev > 1) {
      MHIP(c, hipMalloc(&m->recv0[g], 4 * (size_t)m->n_max * es));
      MHIP(c, hipMalloc(&m->recv1[g], 4 * (size_t)m->n_max * es));
    }
    if (m->R > 0) {
      MHIP(c, hipMalloc(&m->outb[g], 3 * (size_t)m->n_max * es));
      for (int h = 0; h < m->R; ++h) {
        MHIP(c, hipMalloc(&m->back[(size_t)g * m->R + h], 3 * (size_t)m->n_max * es));
        MHIP(c, hipEventCreateWithFlags(&m->ev_back[(size_t)g * m->R + h], hipEventDisableTiming));
      }
    }
    MHIP(c, hipEventCreateWithFlags(&m->ev_added[g], hipEventDisableTiming));
    MHIP(c, hipStreamCreateWithFlags(&m->copy[g], hipStreamNonBlocking));
    MHIP(c, hipEventCreateWithFlags(&m->ev_pos[g], hipEventDisableTiming));
    for (int b = 0; b < 2; ++b) {
      MHIP(c, hipEventCreateWithFlags(&m->ev_copy[g * 2 + b], hipEventDisableTiming));
      MHIP(c, hipEventCreateWithFlags(&m->ev_kern[g * 2 + b], hipEventDisableTiming));
    }
    for (int o = 0; o < ndev; ++o) MHIP(c, hipEventCreateWithFlags(&m->ev_pulled[o * ndev + g], hipEventDisableTiming));
    for (int o = 0; o < ndev; ++o)                       // direct peer reads over xGMI where the devices differ
      if (dev_ids[o] != dev_ids[g]) {
        int can = 0;
        MHIP(c, hipDeviceCanAccessPeer(&can, dev_ids[g], dev_ids[o]));
        if (can) {
          hipError_t e = hipDeviceEnablePeerAccess(dev_ids[o], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return nb_fail(c, NB_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
          (void)hipGetLastError();
        }
      }
  }
  return NB_OK;
}

int nb_multi_set_state(nb_ctx* c, const void* x, const void* y, const void* z, const void* vx, const void* vy, const void* vz, const void* mm) {
  nb_multi_state& m = *c->multi;
  const size_t es = esz(c->dtype);
  int rc = sync_all(c); if (rc) return rc;
  for (int g = 0; g < m.P; ++g) {
    const size_t o = (size_t)m.off[g] * es;
    rc = nb_set_state(m.shard[g], (const char*)x + o, (const char*)y + o, (const char*)z + o, (const char*)vx + o, (const char*)vy + o,
                      (const char*)vz + o, (const char*)mm + o);
    if (rc) return nb_fail(c, rc, m.shard[g]->err);
  }
  bool uni = true;                                       // equal masses over the WHOLE system (every shard alone is not enough)
  if (c->dtype == NB_F32) { const float* q = (const float*)mm; for (int64_t i = 1; i < c->n && uni; ++i) uni = q[i] == q[0]; }
  else { const double* q = (const double*)mm; for (int64_t i = 1; i < c->n && uni; ++i) uni = q[i] == q[0]; }
  m.uniform = uni;
  m.acc_valid = false;
  std::fill(m.pulled_valid.begin(), m.pulled_valid.end(), 0);
  c->has_state = true;
  return NB_OK;
}

int nb_multi_get(nb_ctx* c, void* const* dst, const int* fields, int nf) {
  nb_multi_state& m = *c->multi;
  const size_t es = esz(c->dtype);
  int rc = sync_all(c); if (rc) return rc;
  for (int g = 0; g < m.P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    for (int f = 0; f < nf; ++f)
      if (dst[f]) MHIP(c, hipMemcpyAsync((char*)dst[f] + (size_t)m.off[g] * es, field_ptr(m.shard[g], fields[f]), (size_t)m.cnt[g] * es,
                                         hipMemcpyDeviceToHost, m.shard[g]->stream));
  }
  return sync_all(c);
}

int nb_multi_accel(nb_ctx* c, double eps, double G) {
  return refresh_acc(c, eps, G);
}

bool nb_multi_acc_valid(const nb_ctx* c, double eps, double G) {
  const nb_multi_state& m = *c->multi;
  return m.acc_valid && m.acc_eps == eps && m.acc_G == G;
}

bool nb_multi_has_acc(const nb_ctx* c) { return c->multi->acc_valid; }

int nb_multi_kdk(nb_ctx* c, int64_t nsteps, const double* kick1, const double* drift, const double* kick2, double eps, double G) {
  nb_multi_state& m = *c->multi;
  int rc;
  if (!nb_multi_acc_valid(c, eps, G)) { rc = refresh_acc(c, eps, G); if (rc) return rc; }
  for (int64_t k = 0; k < nsteps; ++k) {
    // closing kick of step k-1 fused with the opening kick and the drift of step k (each rounded separately: nb_launch_kick_drift)
    for (int g = 0; g < m.P; ++g) {
      MHIP(c, hipSetDevice(m.dev[g]));
      nb_ctx* s = m.shard[g];
      rc = wait_for_readers(c, g); if (rc) return rc;
      rc = nb_launch_kick_drift(s, c->dtype, field_ptr(s, 0), field_ptr(s, 1), field_ptr(s, 2), field_ptr(s, 4), field_ptr(s, 5), field_ptr(s, 6),
                                field_ptr(s, 7), field_ptr(s, 8), field_ptr(s, 9), s->n, k == 0 ? 0.0 : kick2[k - 1], kick1[k], drift[k]);
      if (rc) return nb_fail(c, rc, s->err);
    }
    rc = refresh_acc(c, eps, G); if (rc) return rc;
  }
  for (int g = 0; g < m.P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    nb_ctx* s = m.shard[g];
    rc = nb_launch_kick(s, c->dtype, field_ptr(s, 4), field_ptr(s, 5), field_ptr(s, 6), field_ptr(s, 7), field_ptr(s, 8), field_ptr(s, 9), s->n, kick2[nsteps - 1]);
    if (rc) return nb_fail(c, rc, s->err);
  }
  return sync_all(c);
}

int nb_multi_energy(nb_ctx* c, double eps, double G, double* kinetic, double* potential) {
  nb_multi_state& m = *c->multi;
  const size_t es = esz(c->dtype);
  int rc;
  for (int g = 0; g < m.P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    nb_ctx* s = m.shard[g];
    rc = nb_launch_kinetic(s, c->dtype, field_ptr(s, 4), field_ptr(s, 5), field_ptr(s, 6), field_ptr(s, 3), s->n, s->scal);
    if (rc) return nb_fail(c, rc, s->err);
  }
  rc = for_each_chunk(c, [&](int g, int o, const char* src, bool first) -> int {
    nb_ctx* s = m.shard[g];
    const size_t so = (size_t)m.cnt[o] * es;
    if (c->dtype == NB_F32)
      return nb_launch_potential_f32(s, (const float*)field_ptr(s, 0), (const float*)field_ptr(s, 1), (const float*)field_ptr(s, 2), (const float*)field_ptr(s, 3),
                                     s->n, m.off[g], (const float*)src, (const float*)(src + so), (const float*)(src + 2 * so), (const float*)(src + 3 * so),
                                     m.cnt[o], m.off[o], (float)(eps * eps), s->scal + 1, first ? 0 : 1);
    return nb_launch_potential_f64(s, (const double*)field_ptr(s, 0), (const double*)field_ptr(s, 1), (const double*)field_ptr(s, 2), (const double*)field_ptr(s, 3),
                                   s->n, m.off[g], (const double*)src, (const double*)(src + so), (const double*)(src + 2 * so), (const double*)(src + 3 * so),
                                   m.cnt[o], m.off[o], eps * eps, s->scal + 1, first ? 0 : 1);
  });
  if (rc) return rc;
  double K = 0.0, W = 0.0;
  for (int g = 0; g < m.P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    double h[2];
    MHIP(c, hipMemcpyAsync(h, m.shard[g]->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, m.shard[g]->stream));
    MHIP(c, hipStreamSynchronize(m.shard[g]->stream));
    K += h[0]; W += h[1];                                  // fixed order: device 0, 1, ...
  }
  rc = sync_all(c); if (rc) return rc;
  if (kinetic) *kinetic = K;
  if (potential) *potential = -0.5 * G * W;
  return NB_OK;
}

// KDK steps with the energies of the WHOLE system after every step: each device keeps a log of (K, W) partials of its own sinks — the potential
// from the fused force+potential kernels, accumulated over the chunks as they arrive — and the host adds the P logs once, at the end.
int nb_multi_step_energy(nb_ctx* c, double dt, int64_t nsteps, double eps, double G, double* kinetic, double* potential) {
  nb_multi_state& m = *c->multi;
  int rc;
  for (int g = 0; g < m.P; ++g) {
    MHIP(c, hipSetDevice(m.dev[g]));
    nb_ctx* s = m.shard[g];
    if (!s->phi) MHIP(c, hipMalloc((void**)&s->phi, (size_t)s->n * sizeof(double)));
    if (s->elog_cap < nsteps) {
      MHIP(c, hipStreamSynchronize(s->stream));
      if (s->elog) { (void)hipFree(s->elog); s->elog = nullptr; s->elog_cap = 0; }
      MHIP(c, hipMalloc((void**)&s->elog, (size_t)nsteps * 2 * sizeof(double)));
      s->elog_cap = nsteps;
    }
  }
  if (!nb_multi_acc_valid(c, eps, G)) { rc = refresh_acc(c, eps, G); if (rc) return rc; }
  const double h = 0.5 * dt;
  for (int64_t k = 0; k < nsteps; ++k) {
    for (int g = 0; g < m.P; ++g) {
      MHIP(c, hipSetDevice(m.dev[g]));
      nb_ctx* s = m.shard[g];
      rc = wait_for_readers(c, g); if (rc) return rc;
      rc = nb_launch_kick_drift(s, c->dtype, field_ptr(s, 0), field_ptr(s, 1), field_ptr(s, 2), field_ptr(s, 4), field_ptr(s, 5), field_ptr(s, 6),
                                field_ptr(s, 7), field_ptr(s, 8), field_ptr(s, 9), s->n, 0.0, h, dt);
      if (rc) return nb_fail(c, rc, s->err);
    }
    rc = refresh_acc(c, eps, G, true); if (rc) return rc;
    for (int g = 0; g < m.P; ++g) {
      MHIP(c, hipSetDevice(m.dev[g]));
      nb_ctx* s = m.shard[g];
      rc = nb_launch_kick(s, c->dtype, field_ptr(s, 4), field_ptr(s, 5), field_ptr(s, 6), field_ptr(s, 7), field_ptr(s, 8), field_ptr(s, 9), s->n, h);
      if (!rc) rc = nb_launch_kinetic(s, c->dtype, field_ptr(s, 4), field_ptr(s, 5), field_ptr(s, 6), field_ptr(s, 3), s->n, s->elog + 2 * k);
      if (!rc) rc = nb_launch_pot_energy(s, c->dtype, s->phi, field_ptr(s, 3), s->n, eps * eps, c->dtype == NB_F32 && m.uniform, s->elog + 2 * k + 1);
      if (rc) return nb_fail(c, rc, s->err);
    }
  }
  std::vector<double> sum((size_t)nsteps * 2, 0.0), part((size_t)nsteps * 2);
  for (int g = 0; g < m.P; ++g) {                              // fixed order: device 0, 1, ...
    MHIP(c, hipSetDevice(m.dev[g]));
    MHIP(c, hipMemcpyAsync(part.data(), m.shard[g]->elog, part.size() * sizeof(double), hipMemcpyDeviceToHost, m.shard[g]->stream));
    MHIP(c, hipStreamSynchronize(m.shard[g]->stream));
    for (size_t i = 0; i < sum.size(); ++i) sum[i] += part[i];
  }
  rc = sync_all(c); if (rc) return rc;
  for (int64_t k = 0; k < nsteps; ++k) {
    if (kinetic) kinetic[k] = sum[2 * k];
    if (potential) potential[k] = -0.5 * G * sum[2 * k + 1];
  }
  return NB_OK;
}

int nb_multi_sync(nb_ctx* c) { return sync_all(c); }

int nb_multi_info(const nb_ctx* c, int which, int64_t* out, int cap) {
  const nb_multi_state& m = *c->multi;
  for (int g = 0; g < m.P && g < cap; ++g) out[g] = which == 0 ? m.cnt[g] : (which == 1 ? m.off[g] : (int64_t)m.dev[g]);
  return m.P;
}

// options of the pair-symmetric kernels reach every shard; NB_OPT_PAIR_SYMMETRY also selects the schedule of this context
int nb_multi_set_option(nb_ctx* c, int option, int64_t value) {
  nb_multi_state& m = *c->multi;
  int rc = sync_all(c); if (rc) return rc;
  if (option == NB_OPT_PAIR_SYMMETRY) m.pairwise = value < 0 ? -1 : (long long)value;
  for (int g = 0; g < m.P; ++g) {
    rc = nb_set_option(m.shard[g], option, value);
    if (rc) return nb_fail(c, rc, m.shard[g]->err);
  }
  m.acc_valid = false;
  return NB_OK;
}

// nb_last_accel_symmetric of a multi-device context: the OWN-CHUNK launch of device 0 in the last force evaluation (the later launches of
// the one-sided schedule are one-sided by construction; those of the pair-symmetric schedule fall back to two one-sided launches for very
// small chunks — neither says which kernel the own chunk took, so it is recorded when that launch is made)
int nb_multi_last_symmetric(const nb_ctx* c) {
  const nb_multi_state& m = *c->multi;
  return m.own_chunk_sym ? 1 : 0;
}

// NB_QUERY_MULTI_SCHEDULE: 1 = the pair-symmetric schedule is what this context runs (more than one device and the option not switched off)
int nb_multi_schedule_pairwise(const nb_ctx* c) {
  const nb_multi_state& m = *c->multi;
  return (m.P > 1 && m.pairwise != 0) ? 1 : 0;
}
