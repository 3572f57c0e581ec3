/*
 * graspbal_oracle.c — CPU ORACLE for the GraspBalance point-cloud hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this file's library; the product (graspbalance_amd/) never does.
 *
 * It is a scalar restatement, in plain C, of the algorithms of the reference's CUDA kernels
 * (which have no CPU path: PointNet/_ext_src/src/sampling.cpp:39 "CPU not supported").  Each
 * function cites the reference lines it follows (paths relative to the reference tree).
 * Floating point: built with -ffp-contract=off so a*a+b*b+c*c is evaluated as ((a*a)+(b*b))+(c*c)
 * with one rounding per operation — the convention pinned against the reference's importable
 * torch fallback (TrainModel/pointnet2_util.py) by tests/golden/ (see tests/test_oracle_golden.py).
 *
 * Host pointers, same argument order as include/graspbal.h with the prefix gbo_ and no stream.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GBO_FPS_SKIP_NEAR_ORIGIN 0x1u
#define GBO_FPS_TIE_LOWEST 0x00u
#define GBO_FPS_TIE_TREE512 0x10u
#define GBO_FPS_TIE_TREE1024 0x20u
#define GBO_FPS_TIE_MASK 0x30u

/* cuda_utils.h:21-27 opt_n_threads: clamp(2^floor(log2 work_size), 1, cap) */
static int opt_n_threads(int work_size, int cap) {
  int p = 1;
  if (work_size < 1) return 1;
  while ((p << 1) <= work_size && (p << 1) <= cap) p <<= 1;
  return p;
}

static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  /* sampling_gpu.cu:108-109 / ball_query_gpu.cu:30-31 — left-to-right, no contraction */
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return ((dx * dx) + (dy * dy)) + (dz * dz);
}

/* ---------------------------------------------------------------------------------------------
 * FPS.  PN-ext sampling_gpu.cu:75-178 (skip rule :105-106, local scan :113-114, tree :64-70,
 * :119-173), PB-ext pointnet2_batch/src/sampling_gpu.cu:74-181 (no skip, block up to 1024).
 * TIE_LOWEST is the rule of the torch fallback (pointnet2_util.py:41 torch.max -> first index).
 * ------------------------------------------------------------------------------------------- */
int gbo_fps(const float *xyz, float *temp_io, int32_t *idx, int b, int n, int m, unsigned flags) {
  if (b < 0 || n < 1 || m < 0) return -1;
  if (m == 0 || b == 0) return 0;
  const unsigned tie = flags & GBO_FPS_TIE_MASK;
  const int skip = (flags & GBO_FPS_SKIP_NEAR_ORIGIN) != 0;
  int bs = 1;
  if (tie == GBO_FPS_TIE_TREE512) bs = opt_n_threads(n, 512);
  else if (tie == GBO_FPS_TIE_TREE1024) bs = opt_n_threads(n, 1024);
  int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *p = xyz + (size_t)bi * n * 3;
    int32_t *out = idx + (size_t)bi * m;
    float *temp = (float *)malloc(sizeof(float) * (size_t)n);
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    if (!temp || !dists || !dists_i) { rc = -1; free(temp); free(dists); free(dists_i); continue; }
    if (temp_io) memcpy(temp, temp_io + (size_t)bi * n, sizeof(float) * (size_t)n);
    else for (int k = 0; k < n; ++k) temp[k] = 1e10f; /* sampling.cpp:78-80 */
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
      if (tie == GBO_FPS_TIE_LOWEST) {
        int besti = 0;
        float best = -1.0f;
        for (int k = 0; k < n; ++k) {
          const float x2 = p[k * 3 + 0], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
          if (skip) {
            const float mag = ((x2 * x2) + (y2 * y2)) + (z2 * z2);
            if (mag <= 1e-3) continue; /* double compare like the .cu (1e-3 is a double literal) */
          }
          const float d = sqdist3(x2, y2, z2, x1, y1, z1);
          const float d2 = fminf(d, temp[k]);
          temp[k] = d2;
          if (d2 > best) { best = d2; besti = k; }
        }
        old = besti;
      } else {
        for (int t = 0; t < bs; ++t) {
          int besti = 0;
          float best = -1.0f;
          for (int k = t; k < n; k += bs) {
            const float x2 = p[k * 3 + 0], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
            if (skip) {
              const float mag = ((x2 * x2) + (y2 * y2)) + (z2 * z2);
              if (mag <= 1e-3) continue;
            }
            const float d = sqdist3(x2, y2, z2, x1, y1, z1);
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            besti = d2 > best ? k : besti;
            best = d2 > best ? d2 : best;
          }
          dists[t] = best;
          dists_i[t] = besti;
        }
        for (int s = bs >> 1; s >= 1; s >>= 1) { /* __update, sampling_gpu.cu:64-70 */
          for (int t = 0; t < s; ++t) {
            const float v1 = dists[t], v2 = dists[t + s];
            const int i1 = dists_i[t], i2 = dists_i[t + s];
            dists[t] = v1 > v2 ? v1 : v2;
            dists_i[t] = v2 > v1 ? i2 : i1;
          }
        }
        old = dists_i[0];
      }
      out[j] = old;
    }
    if (temp_io) memcpy(temp_io + (size_t)bi * n, temp, sizeof(float) * (size_t)n);
    free(temp); free(dists); free(dists_i);
  }
  return rc;
}

/* sampling_gpu.cu:13-25 */
int gbo_gather(const float *points, const int32_t *idx, float *out, int b, int c, int n, int m) {
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]];
  return 0;
}

/* sampling_gpu.cu:39-52 (serial order j = 0..m-1 per (b,c) row; the GPU order is undefined) */
int gbo_gather_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c,
                    int n, int m) {
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        grad_points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]] +=
            grad_out[((size_t)i * c + l) * m + j];
  return 0;
}

/* ball_query_gpu.cu:9-44 (PB-ext ball_query_gpu.cu:10-42 is the same predicate and order).
 * scanned[b,m] (optional) = number of points the serial scan visits before it stops. */
int gbo_ball_query(const float *new_xyz, const float *xyz, int32_t *idx, int32_t *scanned, int b,
                   int n, int m, float radius, int nsample) {
  const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static, 16)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *q = new_xyz + ((size_t)bi * m + j) * 3;
      const float *p = xyz + (size_t)bi * n * 3;
      int32_t *row = idx + ((size_t)bi * m + j) * nsample;
      for (int l = 0; l < nsample; ++l) row[l] = 0; /* torch::zeros, ball_query.cpp:24-26 */
      const float nx = q[0], ny = q[1], nz = q[2];
      int cnt = 0, k = 0;
      for (; k < n && cnt < nsample; ++k) {
        const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const float d2 = sqdist3(nx, ny, nz, x, y, z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) row[l] = k;
          row[cnt] = k;
          ++cnt;
        }
      }
      if (scanned) scanned[(size_t)bi * m + j] = k;
    }
  return 0;
}

/* cylinder_query_gpu.cu:20-78 */
int gbo_cylinder_query(const float *new_xyz, const float *xyz, const float *rot, int32_t *idx,
                       int32_t *scanned, int b, int n, int m, float radius, float hmin, float hmax,
                       int nsample) {
  const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static, 16)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *q = new_xyz + ((size_t)bi * m + j) * 3;
      const float *r = rot + ((size_t)bi * m + j) * 9;
      const float *p = xyz + (size_t)bi * n * 3;
      int32_t *row = idx + ((size_t)bi * m + j) * nsample;
      for (int l = 0; l < nsample; ++l) row[l] = 0;
      const float nx = q[0], ny = q[1], nz = q[2];
      int cnt = 0, k = 0;
      for (; k < n && cnt < nsample; ++k) {
        const float x = p[k * 3 + 0] - nx, y = p[k * 3 + 1] - ny, z = p[k * 3 + 2] - nz;
        const float x_rot = ((r[0] * x) + (r[3] * y)) + (r[6] * z); /* :62-64 */
        const float y_rot = ((r[1] * x) + (r[4] * y)) + (r[7] * z);
        const float z_rot = ((r[2] * x) + (r[5] * y)) + (r[8] * z);
        const float d2 = (y_rot * y_rot) + (z_rot * z_rot);
        if (d2 < radius2 && x_rot > hmin && x_rot < hmax) { /* :66 */
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) row[l] = k;
          row[cnt] = k;
          ++cnt;
        }
      }
      if (scanned) scanned[(size_t)bi * m + j] = k;
    }
  return 0;
}

/* group_points_gpu.cu:17-44 */
int gbo_group(const float *points, const int32_t *idx, float *out, int b, int c, int n, int m,
              int nsample) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *src = points + ((size_t)bi * c + l) * n;
      const int32_t *ix = idx + (size_t)bi * m * nsample;
      float *dst = out + ((size_t)bi * c + l) * m * nsample;
      for (size_t e = 0; e < (size_t)m * nsample; ++e) dst[e] = src[ix[e]];
    }
  return 0;
}

/* group_points_gpu.cu:69-90 (serial (j,k) order per (b,c) row) */
int gbo_group_grad(const float *grad_out, const int32_t *idx, float *grad_points, int b, int c,
                   int n, int m, int nsample) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *dst = grad_points + ((size_t)bi * c + l) * n;
      const int32_t *ix = idx + (size_t)bi * m * nsample;
      const float *src = grad_out + ((size_t)bi * c + l) * m * nsample;
      for (size_t e = 0; e < (size_t)m * nsample; ++e) dst[ix[e]] += src[e];
    }
  return 0;
}

/* interpolate_gpu.cu:14-64: bests kept in double, compared against the float distance */
int gbo_three_nn(const float *unknown, const float *known, float *dist2, int32_t *idx, int b, int n,
                 int m) {
#pragma omp parallel for collapse(2) schedule(static, 64)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const float *u = unknown + ((size_t)bi * n + j) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      const float ux = u[0], uy = u[1], uz = u[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = sqdist3(ux, uy, uz, kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float *od = dist2 + ((size_t)bi * n + j) * 3;
      int32_t *oi = idx + ((size_t)bi * n + j) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
  return 0;
}

/* interpolate_gpu.cu:77-106 */
int gbo_three_interpolate(const float *points, const int32_t *idx, const float *weight, float *out,
                          int b, int c, int m, int n) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *src = points + ((size_t)bi * c + l) * m;
      float *dst = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float *w = weight + ((size_t)bi * n + j) * 3;
        const int32_t *ix = idx + ((size_t)bi * n + j) * 3;
        dst[j] = ((src[ix[0]] * w[0]) + (src[ix[1]] * w[1])) + (src[ix[2]] * w[2]);
      }
    }
  return 0;
}

/* interpolate_gpu.cu:121-148 */
int gbo_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                               float *grad_points, int b, int c, int n, int m) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *dst = grad_points + ((size_t)bi * c + l) * m;
      const float *src = grad_out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float *w = weight + ((size_t)bi * n + j) * 3;
        const int32_t *ix = idx + ((size_t)bi * n + j) * 3;
        dst[ix[0]] += src[j] * w[0];
        dst[ix[1]] += src[j] * w[1];
        dst[ix[2]] += src[j] * w[2];
      }
    }
  return 0;
}

/* KNN/Pytorch_CUDA_KNN/cpu/knn_cpu.cpp:4-55 for any k <= nref: squared distances accumulated over the `dim` rows in order,
 * then the k smallest in ascending order with equal distances in index order (what the reference's stable bubble sort of
 * the whole row leaves in its first k places).  idx (b,k,nq), 1-based. */
int gbo_knn(const float *ref, const float *query, int64_t *idx, int b, int dim, int nref, int nq, int k) {
  if (k < 1 || k > nref) return -1;
#pragma omp parallel for collapse(2) schedule(static, 64)
  for (int bi = 0; bi < b; ++bi)
    for (int q = 0; q < nq; ++q) {
      const float *r = ref + (size_t)bi * dim * nref;
      const float *qq = query + (size_t)bi * dim * nq;
      float bd[64];
      int64_t bi_[64];
      int have = 0;
      for (int c = 0; c < nref; ++c) {
        float d = 0.0f;
        for (int h = 0; h < dim; ++h) {
          const float t = r[(size_t)h * nref + c] - qq[(size_t)h * nq + q];
          d += t * t;
        }
        int pos = have;                       /* after every entry that is <= d: stable */
        while (pos > 0 && d < bd[pos - 1]) --pos;
        if (pos >= k) continue;
        const int last = have < k ? have : k - 1;
        for (int j = last; j > pos; --j) { bd[j] = bd[j - 1]; bi_[j] = bi_[j - 1]; }
        bd[pos] = d;
        bi_[pos] = c + 1;
        if (have < k) ++have;
      }
      for (int i = 0; i < k; ++i) idx[((size_t)bi * k + i) * nq + q] = bi_[i];
    }
  return 0;
}

/* KNN/Pytorch_CUDA_KNN/cpu/knn_cpu.cpp:4-55 with k = 1: squared distance accumulated over the
 * `dim` rows in order, stable bubble sort => lowest index among equal minima, 1-based output. */
int gbo_knn1(const float *ref, const float *query, int64_t *idx, int b, int dim, int nref, int nq) {
#pragma omp parallel for collapse(2) schedule(static, 64)
  for (int bi = 0; bi < b; ++bi)
    for (int q = 0; q < nq; ++q) {
      const float *r = ref + (size_t)bi * dim * nref;
      const float *qq = query + (size_t)bi * dim * nq;
      float best = INFINITY;
      int64_t besti = 1;
      for (int k = 0; k < nref; ++k) {
        float d = 0.0f;
        for (int h = 0; h < dim; ++h) {
          const float t = r[(size_t)h * nref + k] - qq[(size_t)h * nq + q];
          d += t * t;
        }
        if (k == 0 || d < best) { best = d; besti = k + 1; }
      }
      idx[(size_t)bi * nq + q] = besti;
    }
  return 0;
}
