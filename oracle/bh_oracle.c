/* bh_oracle.c — sequential CPU restatement of the Barnes-Hut algorithm of bachelor_thesis_n_body_amd/csrc/bh.hip.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ may load libbh_oracle.so).
 * PARITY UNPINNED against the reference: the reference has no tree code at all; BASELINE.json
 * configs[4] only names "Barnes-Hut theta = 0.5". The algorithm is therefore DEFINED by this project; this file states
 * it step by step in plain C so that the HIP kernels can be checked against an independent, sequential execution of
 * the same definition, and both are checked against direct summation (tests/test_gpu_bh.py, tests/test_bh_oracle.py):
 *   1. bounding cube (fp32 min/max; edge = max extent * 1.0001f, 1 if degenerate)
 *   2. 63-bit Morton keys, 21 bits per axis: cell = (unsigned)clamp((v - lo) * (2^21 / edge), 0, 2^21 - 1)
 *   3. stable sort by (key, original index)
 *   4. binary radix tree over the sorted keys (Karras 2012), delta(i,j) = clz(key_i ^ key_j), ties broken by position
 *   5. node mass / centre of mass from fp64 inclusive prefix sums of m, m x, m y, m z in sorted order, rounded to fp32;
 *      node size l = edge * 2^-level, level = (common prefix bits - 1 unused top bit) / 3, 21 if all keys are equal
 *   6. walk per group of 64 consecutive sorted particles, depth first (left child first): accept a node when
 *      l^2 < theta^2 d^2 for EVERY particle of the group (d to the node's centre of mass); a rejected node with at most
 *      16 particles is summed directly; otherwise its two children are visited.
 * fp32 arithmetic as in the kernel (fmaf accumulation); rsqrt is 1/sqrtf here and v_rsq_f32 (1 ulp) on the GPU.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NLEAF 16
#define NLEAF_LIST 4   /* bh.hip: NLEAF_LIST */

typedef struct { uint64_t key; int idx; } kv_t;

static int cmp_kv(const void* a, const void* b) {
  const kv_t *x = (const kv_t*)a, *y = (const kv_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

static uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

static int delta(const kv_t* k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  if (k[i].key != k[j].key) return __builtin_clzll(k[i].key ^ k[j].key);
  return 64 + __builtin_clz((unsigned)i ^ (unsigned)j);
}

typedef struct {
  int n;
  float edge;
  kv_t* kv;
  float *sx, *sy, *sz, *sm;          /* sorted particles */
  int *left, *right, *first, *count; /* nodes: internal 0..n-2, leaves n-1..2n-2 */
  float *cx, *cy, *cz, *cm, *l2;
} tree_t;

static void tree_free(tree_t* t) {
  free(t->kv); free(t->sx); free(t->sy); free(t->sz); free(t->sm); free(t->left); free(t->right);
  free(t->first); free(t->count); free(t->cx); free(t->cy); free(t->cz); free(t->cm); free(t->l2);
}

static void tree_build(tree_t* t, int n, const float* x, const float* y, const float* z, const float* m) {
  memset(t, 0, sizeof(*t));
  t->n = n;
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = 0; i < n; ++i) {
    const float v[3] = {x[i], y[i], z[i]};
    for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], v[c]); hi[c] = fmaxf(hi[c], v[c]); }
  }
  float edge = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
  edge = edge > 0.f ? edge * 1.0001f : 1.0f;
  t->edge = edge;
  const float scale = 2097152.0f / edge;
  t->kv = (kv_t*)malloc(sizeof(kv_t) * n);
  for (int i = 0; i < n; ++i) {
    const float v[3] = {x[i], y[i], z[i]};
    uint32_t q[3];
    for (int c = 0; c < 3; ++c) {
      float tt = (v[c] - lo[c]) * scale;
      tt = fminf(fmaxf(tt, 0.0f), 2097151.0f);
      q[c] = (uint32_t)tt;
    }
    t->kv[i].key = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    t->kv[i].idx = i;
  }
  qsort(t->kv, n, sizeof(kv_t), cmp_kv);
  t->sx = malloc(4 * (size_t)n); t->sy = malloc(4 * (size_t)n); t->sz = malloc(4 * (size_t)n); t->sm = malloc(4 * (size_t)n);
  double* pre = (double*)malloc(sizeof(double) * 4 * (size_t)n);
  double acc[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const int j = t->kv[i].idx;
    t->sx[i] = x[j]; t->sy[i] = y[j]; t->sz[i] = z[j]; t->sm[i] = m[j];
    const double mm = (double)m[j];
    acc[0] += mm; acc[1] += mm * (double)x[j]; acc[2] += mm * (double)y[j]; acc[3] += mm * (double)z[j];
    for (int c = 0; c < 4; ++c) pre[(size_t)c * n + i] = acc[c];
  }
  const size_t total = 2 * (size_t)n - 1;
  t->left = malloc(4 * total); t->right = malloc(4 * total); t->first = malloc(4 * total); t->count = malloc(4 * total);
  t->cx = malloc(4 * total); t->cy = malloc(4 * total); t->cz = malloc(4 * total); t->cm = malloc(4 * total);
  t->l2 = malloc(4 * total);
  const kv_t* k = t->kv;
  for (int i = 0; i < n - 1; ++i) {
    const int d = (delta(k, n, i, i + 1) - delta(k, n, i, i - 1) >= 0) ? 1 : -1;
    const int dmin = delta(k, n, i, i - d);
    int lmax = 2;
    while (delta(k, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int tt = lmax >> 1; tt >= 1; tt >>= 1)
      if (delta(k, n, i, i + (l + tt) * d) > dmin) l += tt;
    const int j = i + l * d;
    const int dnode = delta(k, n, i, j);
    int s = 0;
    for (int tt = (l + 1) >> 1;; tt = (tt + 1) >> 1) {
      if (delta(k, n, i, i + (s + tt) * d) > dnode) s += tt;
      if (tt == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    t->left[i] = (first == gamma) ? (n - 1 + gamma) : gamma;
    t->right[i] = (last == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
    t->first[i] = first; t->count[i] = last - first + 1;
    const int pbits = dnode >= 64 ? 63 : (dnode - 1 > 0 ? dnode - 1 : 0);
    const float cell = ldexpf(edge, -(pbits / 3));
    t->l2[i] = cell * cell;
    double s4[4];
    for (int c = 0; c < 4; ++c) s4[c] = pre[(size_t)c * n + last] - (first > 0 ? pre[(size_t)c * n + first - 1] : 0.0);
    if (s4[0] > 0.0) { t->cx[i] = (float)(s4[1] / s4[0]); t->cy[i] = (float)(s4[2] / s4[0]); t->cz[i] = (float)(s4[3] / s4[0]); t->cm[i] = (float)s4[0]; }
    else { t->cx[i] = t->sx[first]; t->cy[i] = t->sy[first]; t->cz[i] = t->sz[first]; t->cm[i] = 0.0f; }
  }
  for (int j = 0; j < n; ++j) {
    const int v = n - 1 + j;
    t->left[v] = t->right[v] = -1; t->first[v] = j; t->count[v] = 1;
    t->cx[v] = t->sx[j]; t->cy[v] = t->sy[j]; t->cz[v] = t->sz[j]; t->cm[v] = t->sm[j]; t->l2[v] = 0.0f;
  }
  free(pre);
}

/* Accelerations of all particles (original order). stats[0] = node-monopole interactions, stats[1] = direct
 * particle interactions, stats[2] = nodes visited (summed over groups), may be NULL. Returns 0, or -1 if n < 128
 * (the product falls back to the direct sum there and so should the caller). */
int bho_accel_f32(int n, const float* x, const float* y, const float* z, const float* m, float eps, float G,
                  float theta, float* ax, float* ay, float* az, int64_t* stats) {
  if (n < 128 || !(theta > 0.f)) return -1;
  tree_t t;
  tree_build(&t, n, x, y, z, m);
  const float eps2 = eps * eps, theta2 = theta * theta;
  int64_t st[3] = {0, 0, 0};
  int* stack = (int*)malloc(sizeof(int) * 4096);
  for (int g0 = 0; g0 < n; g0 += 64) {
    const int g1 = g0 + 64 < n ? g0 + 64 : n;
    float fx[64], fy[64], fz[64];
    for (int q = 0; q < 64; ++q) fx[q] = fy[q] = fz[q] = 0.f;
    int top = 0;
    stack[top++] = 0;
    while (top > 0) {
      const int v = stack[--top];
      ++st[2];
      const int is_leaf = v >= n - 1;
      int all_ok = !is_leaf;
      if (!is_leaf)
        for (int i = g0; i < g1; ++i) {
          const float dx = t.cx[v] - t.sx[i], dy = t.cy[v] - t.sy[i], dz = t.cz[v] - t.sz[i];
          const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
          if (!(t.l2[v] < theta2 * d2)) { all_ok = 0; break; }
        }
      if (all_ok) {
        for (int i = g0; i < g1; ++i) {
          const float dx = t.cx[v] - t.sx[i], dy = t.cy[v] - t.sy[i], dz = t.cz[v] - t.sz[i];
          const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz)) + eps2;
          const float rinv = r2 > 0.f ? 1.0f / sqrtf(r2) : 0.f;
          const float f = t.cm[v] * rinv * rinv * rinv;
          fx[i - g0] = fmaf(f, dx, fx[i - g0]); fy[i - g0] = fmaf(f, dy, fy[i - g0]); fz[i - g0] = fmaf(f, dz, fz[i - g0]);
        }
        st[0] += g1 - g0;
      } else if (is_leaf || t.count[v] <= NLEAF) {
        for (int j = t.first[v]; j < t.first[v] + t.count[v]; ++j)
          for (int i = g0; i < g1; ++i) {
            const float ex = t.sx[j] - t.sx[i], ey = t.sy[j] - t.sy[i], ez = t.sz[j] - t.sz[i];
            const float r2 = fmaf(ex, ex, fmaf(ey, ey, ez * ez)) + eps2;
            const float rinv = r2 > 0.f ? 1.0f / sqrtf(r2) : 0.f;
            const float f = t.sm[j] * rinv * rinv * rinv;
            fx[i - g0] = fmaf(f, ex, fx[i - g0]); fy[i - g0] = fmaf(f, ey, fy[i - g0]); fz[i - g0] = fmaf(f, ez, fz[i - g0]);
          }
        st[1] += (int64_t)t.count[v] * (g1 - g0);
      } else {
        stack[top++] = t.right[v];
        stack[top++] = t.left[v];
      }
    }
    for (int i = g0; i < g1; ++i) {
      const int j = t.kv[i].idx;
      ax[j] = G * fx[i - g0]; ay[j] = G * fy[i - g0]; az[j] = G * fz[i - g0];
    }
  }
  if (stats) { stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; }
  free(stack);
  tree_free(&t);
  return 0;
}

/* The LIST walk of bh.hip (walk_list_kernel), executed sequentially with the same pass structure: every pass pops up to
 * 64 nodes from the top of the stack ("lane" L takes stack[top - k + L]); a node is accepted when l^2 < theta^2 d^2 with d^2
 * the squared distance from the node's centre of mass to the NEAREST acceptance volume of the group: the bounding boxes of its 8 runs of 8
 * consecutive particles, a run whose box diagonal exceeds 0.6 of the group's entered particle by particle (a lower bound of the distance
 * to every particle: the criterion holds for EVERY particle of the group); accepted nodes form the pass's monopole list
 * (lane order), rejected nodes with <= NLEAF particles its leaf list, the others push (right, left) in lane order. Each
 * particle of the group then adds the monopoles in list order and the leaf particles in list order. */
int bho_accel_list_f32(int n, const float* x, const float* y, const float* z, const float* m, float eps, float G,
                       float theta, float* ax, float* ay, float* az, int64_t* stats) {
  if (n < 128 || !(theta > 0.f)) return -1;
  tree_t t;
  tree_build(&t, n, x, y, z, m);
  const float eps2 = eps * eps, theta2 = theta * theta;
  int64_t st[4] = {0, 0, 0, 0};
  const int cap = 1 << 20;
  int* stack = (int*)malloc(sizeof(int) * cap);
  for (int g0 = 0; g0 < n; g0 += 64) {
    const int g1 = g0 + 64 < n ? g0 + 64 : n;
    float fx[64], fy[64], fz[64];
    for (int q = 0; q < 64; ++q) fx[q] = fy[q] = fz[q] = 0.f;
    /* acceptance volumes (same arithmetic as bh.hip: min / max are exact, the diagonals are the same fmaf chains) */
    float vlo[64][3], vhi[64][3];
    int nvol = 0;
    {
      float gl[3] = {3.4e38f, 3.4e38f, 3.4e38f}, gh[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
      for (int i = g0; i < g1; ++i) {
        const float p[3] = {t.sx[i], t.sy[i], t.sz[i]};
        for (int c = 0; c < 3; ++c) { gl[c] = fminf(gl[c], p[c]); gh[c] = fmaxf(gh[c], p[c]); }
      }
      const float gx = gh[0] - gl[0], gy = gh[1] - gl[1], gz = gh[2] - gl[2];
      const float gdiag2 = fmaf(gx, gx, fmaf(gy, gy, gz * gz));
      for (int b0 = g0; b0 < g1; b0 += 8) {
        const int b1 = b0 + 8 < g1 ? b0 + 8 : g1;
        float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int i = b0; i < b1; ++i) {
          const float p[3] = {t.sx[i], t.sy[i], t.sz[i]};
          for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], p[c]); hi[c] = fmaxf(hi[c], p[c]); }
        }
        const float bx = hi[0] - lo[0], by = hi[1] - lo[1], bz = hi[2] - lo[2];
        const float diag2 = fmaf(bx, bx, fmaf(by, by, bz * bz));
        if (diag2 > 0.36f * gdiag2) {
          for (int i = b0; i < b1; ++i) {
            vlo[nvol][0] = vhi[nvol][0] = t.sx[i]; vlo[nvol][1] = vhi[nvol][1] = t.sy[i]; vlo[nvol][2] = vhi[nvol][2] = t.sz[i]; ++nvol;
          }
        } else {
          for (int c = 0; c < 3; ++c) { vlo[nvol][c] = lo[c]; vhi[nvol][c] = hi[c]; }
          ++nvol;
        }
      }
    }
    int top = 0;
    stack[top++] = 0;
    while (top > 0) {
      const int k = top < 64 ? top : 64;
      top -= k;
      int popped[64], mono[64], leaf[64], nm = 0, nl = 0, no = 0, opened[64];
      for (int L = 0; L < k; ++L) popped[L] = stack[top + L];
      if (top + k > st[3]) st[3] = top + k;
      for (int L = 0; L < k; ++L) {
        const int v = popped[L];
        ++st[2];
        const int is_leaf = v >= n - 1;
        float d2 = 3.4e38f;
        for (int e = 0; e < nvol; ++e) {
          const float ex = fmaxf(fmaxf(vlo[e][0] - t.cx[v], t.cx[v] - vhi[e][0]), 0.f), ey = fmaxf(fmaxf(vlo[e][1] - t.cy[v], t.cy[v] - vhi[e][1]), 0.f),
                      ez = fmaxf(fmaxf(vlo[e][2] - t.cz[v], t.cz[v] - vhi[e][2]), 0.f);
          d2 = fminf(d2, fmaf(ex, ex, fmaf(ey, ey, ez * ez)));
        }
        if (!is_leaf && t.l2[v] < theta2 * d2) mono[nm++] = v;
        else if (is_leaf || t.count[v] <= NLEAF_LIST) leaf[nl++] = v;
        else opened[no++] = v;
      }
      for (int p = 0; p < no; ++p) { stack[top + 2 * p] = t.right[opened[p]]; stack[top + 2 * p + 1] = t.left[opened[p]]; }
      top += 2 * no;
      for (int i = g0; i < g1; ++i) {
        float a0 = fx[i - g0], a1 = fy[i - g0], a2 = fz[i - g0];
        for (int e = 0; e < nm; ++e) {
          const int v = mono[e];
          const float dx = t.cx[v] - t.sx[i], dy = t.cy[v] - t.sy[i], dz = t.cz[v] - t.sz[i];
          const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
          const float rinv = r2 > 0.f ? 1.0f / sqrtf(r2) : 0.f;
          const float f = t.cm[v] * rinv * rinv * rinv;
          a0 = fmaf(f, dx, a0); a1 = fmaf(f, dy, a1); a2 = fmaf(f, dz, a2);
        }
        for (int e = 0; e < nl; ++e) {
          const int v = leaf[e];
          for (int j = t.first[v]; j < t.first[v] + t.count[v]; ++j) {
            const float dx = t.sx[j] - t.sx[i], dy = t.sy[j] - t.sy[i], dz = t.sz[j] - t.sz[i];
            const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
            const float rinv = r2 > 0.f ? 1.0f / sqrtf(r2) : 0.f;
            const float f = t.sm[j] * rinv * rinv * rinv;
            a0 = fmaf(f, dx, a0); a1 = fmaf(f, dy, a1); a2 = fmaf(f, dz, a2);
          }
        }
        fx[i - g0] = a0; fy[i - g0] = a1; fz[i - g0] = a2;
      }
      st[0] += (int64_t)nm * (g1 - g0);
      for (int e = 0; e < nl; ++e) st[1] += (int64_t)t.count[leaf[e]] * (g1 - g0);
    }
    for (int i = g0; i < g1; ++i) {
      const int j = t.kv[i].idx;
      ax[j] = G * fx[i - g0]; ay[j] = G * fy[i - g0]; az[j] = G * fz[i - g0];
    }
  }
  if (stats) { stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; stats[3] = st[3]; }
  free(stack);
  tree_free(&t);
  return 0;
}

/* Tree invariants for tests: returns 0 if every internal node's children partition its range, the root covers
 * [0, n), node masses equal the sum of their leaves (to 1e-6), else a positive error code. */
int bho_check_tree(int n, const float* x, const float* y, const float* z, const float* m) {
  if (n < 2) return 0;
  tree_t t;
  tree_build(&t, n, x, y, z, m);
  int rc = 0;
  if (t.first[0] != 0 || t.count[0] != n) rc = 1;
  for (int i = 0; i < n - 1 && !rc; ++i) {
    const int L = t.left[i], R = t.right[i];
    if (t.first[L] != t.first[i]) rc = 2;
    else if (t.first[L] + t.count[L] != t.first[R]) rc = 3;
    else if (t.count[L] + t.count[R] != t.count[i]) rc = 4;
    else if (t.count[L] < 1 || t.count[R] < 1) rc = 5;
    else if (fabsf(t.cm[L] + t.cm[R] - t.cm[i]) > 1e-5f * fabsf(t.cm[i]) + 1e-30f) rc = 6;
    else if (i > 0 && !(t.l2[i] <= t.l2[0])) rc = 7;
  }
  for (int i = 1; i < n && !rc; ++i) if (cmp_kv(&t.kv[i - 1], &t.kv[i]) >= 0) rc = 8;
  tree_free(&t);
  return rc;
}
