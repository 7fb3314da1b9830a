// Level 2: B scenes advanced in lock step through K inserts, everything resident in HBM.
//
// The algorithm is the incremental one stated in tests/incremental_model.py and DESIGN.md par.3:
// project every scene once (step 0), then per insert evaluate the sample only on its candidate
// pixels of the range image, rebuilt for that window from the living points (r3d_insert.hip); dead
// points are dropped once, in r3d_batch_finish.  This file: step 0 (bounds, projection), the
// compaction and the host entry points around them.
//
// Kernels that walk scenes take a (list, count) pair: block row `blockIdx.y` handles scenes
// list[blockIdx.y], list[blockIdx.y + gridDim.y], ... below *count.
#include "r3d_batch.hpp"

#include <cstdlib>

namespace r3d {

// ---- step 0 / rebase: bounds ------------------------------------------------------------------
__global__ void k_begin_init(r3d_batch_t b, const int32_t *n_points, BatchWs w) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {
    *w.all_count = b.B;
  }
  if (s >= b.B) return;
  int n = n_points[s];
  int st = 0;
  if (n < 0 || n > b.cap) {
    n = 0;
    st = R3D_S_CAPACITY;
  }
  b.n_head[s] = n;
  b.n_total[s] = n;
  b.n_log[s] = 0;
  b.n_far[s] = 0;
  b.rebase[s] = 0;
  b.status[s] = st;
  b.n_out[s] = 0;
  w.all_list[s] = s;
  w.qkeys[2 * s + 0] = ~0ull;   // running min of z/r
  w.qkeys[2 * s + 1] = 0ull;    // running max of z/r
}

// elevation = acos(z/r) is monotone in q = z/r, so the bounds of insertion.py:78-79 are acos of
// the extreme q: reduce q here (sqrt + divide per point), take acos twice per scene afterwards.
__global__ void __launch_bounds__(kPT)
k_bounds(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w) {
  __shared__ unsigned long long s_min[kPT / 64], s_max[kPT / 64];
  int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    int t0 = blockIdx.x * kTile;
    if (t0 >= n) continue;
    unsigned long long lmin = ~0ull, lmax = 0ull;
    int bad = 0;
    // the thread's 8 float32 points are requested together (inserted float64 points, which only exist when a
    // re-based scene comes through here, are fetched from the log below)
    const float4 *src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    float4 pt[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      pt[k] = src[i < n ? i : n - 1];
    }
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      if (i < n) {
        double x = (double)pt[k].x, y = (double)pt[k].y, z = (double)pt[k].z;
        if (i >= n_head) load_point(b, s, i, n_head, x, y, z);
        double r = sqrt(x * x + y * y + z * z);
        double q = z / r;
        if (!(q >= -1.0 && q <= 1.0) || !isfinite(x) || !isfinite(y)) {
          bad = 1;
        } else {
          unsigned long long kq = ordered_key(q);
          lmin = kq < lmin ? kq : lmin;
          lmax = kq > lmax ? kq : lmax;
        }
      }
    }
    lmin = wave_min_u64(lmin);
    lmax = wave_max_u64(lmax);
    bad = wave_or_i32(bad);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      s_min[wave] = lmin;
      s_max[wave] = lmax;
      if (bad) atomicOr(&b.status[s], R3D_S_NONFINITE);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int v = 1; v < kPT / 64; ++v) {
        lmin = s_min[v] < lmin ? s_min[v] : lmin;
        lmax = s_max[v] > lmax ? s_max[v] : lmax;
      }
      atomicMin(&w.qkeys[2 * s + 0], lmin);
      atomicMax(&w.qkeys[2 * s + 1], lmax);
    }
    __syncthreads();
  }
}

// After k_bounds, one block per scene: the elevation bounds (insertion.py:78-79) = acos of the extreme
// z/r, the row-edge table of the verified fast projection (k_project) and the living-point count of
// every compaction tile (all points of the frame are alive at step 0).
//
// Tables of the fast projection: a bin guessed in float32 is accepted only if the point lies
// strictly inside that bin's edges, tested in float64 on monotone images of the edges -- cos of
// the row edges against z/r, and the sign of the cross product with the unit vector of the column
// edges -- with a margin far above the rounding of either side.
__global__ void __launch_bounds__(kPT)
k_prepare(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles) {
  __shared__ double s_b[2];
  int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    if (threadIdx.x == 0) {
      unsigned long long kmin = w.qkeys[2 * s + 0], kmax = w.qkeys[2 * s + 1];
      if (kmin == ~0ull) {                    // no valid point: the reference raises (insertion.py:78)
        b.bounds[2 * s + 0] = b.bounds[2 * s + 1] = s_b[0] = s_b[1] = 0.0;
        atomicOr(&b.status[s], R3D_S_NONFINITE);
      } else {
        double q_lo = ordered_key_inv(kmin), q_hi = ordered_key_inv(kmax);
        s_b[0] = b.bounds[2 * s + 0] = acos(q_lo);     // max elevation, insertion.py:79
        s_b[1] = b.bounds[2 * s + 1] = acos(q_hi);     // min elevation, insertion.py:78
        w.q_ext[2 * s + 0] = q_lo;
        w.q_ext[2 * s + 1] = q_hi;
      }
      b.n_far[s] = 0;
      w.n_slow[s] = 0;
    }
    __syncthreads();
    double max_el = s_b[0], min_el = s_b[1];
    double d_el = (max_el - min_el) / (double)b.rows;
    for (int k = threadIdx.x; k < b.rows + 2; k += kPT) {   // entry k holds edge k-1
      double edge = min_el + 0.00001 + (double)(k - 1) * d_el;
      // outside [0, pi] the cosine stops being monotone: clamp (such rows can hold no point anyway)
      edge = edge < 0.0 ? 0.0 : (edge > kPi ? kPi : edge);
      double c = cos(edge);
      w.row_q[(int64_t)s * (b.rows + 2) + k] = c * fabs(c);   // compared with z*|z| / (x*x+y*y+z*z)
    }
    const int n = b.n_total[s];
    for (int t = threadIdx.x; t < tiles; t += kPT) {
      int left = n - t * kTile;
      w.tile_alive[(int64_t)s * tiles + t] = left < 0 ? 0 : (left > kTile ? kTile : left);
    }
    __syncthreads();
  }
}

__global__ void k_col_table(r3d_batch_t b, BatchWs w) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > b.cols) return;
  double alpha = (double)c * (kTwoPi / (double)b.cols) - kPi;   // direction angle of column edge c
  w.col_dir[2 * c + 0] = cos(alpha);
  w.col_dir[2 * c + 1] = sin(alpha);
}

// ---- step 0 / rebase: spherical projection -> pixel ids ------------------------------------------
// insertion.py:74-76 and :104-116 fused: r/az/el are never stored, only the pixel id of every
// point.  The min-reduce of :118-125 is NOT done here for the whole image: a scene's range image
// is only ever read in the window around an inserted object, so k_insert builds exactly that
// window from the points.  To find those points without scanning the cloud,
// every 64 consecutive points (one wave) leave their row / column bounding box; LiDAR files are
// ring-ordered, so a box is about one row by 50 columns.
// Verified float32 guess of (row, col); returns false when the float64 check cannot confirm the
// guessed bin (the caller then queues the point for the reference formula).
//   rows: elevation in [edge_k, edge_k+1)  <=>  cos(edge_k+1) < z/r <= cos(edge_k); both sides are
//         mapped through t -> t*|t| (strictly increasing) and multiplied by r*r = ss, which needs
//         neither the square root nor the division: z*|z| against c*|c| * ss.  Row 0 also takes
//         the truncated interval below edge_0 (int() rounds toward zero).
//   cols: the point lies counter-clockwise of column edge k and clockwise of edge k+1 (sign of
//         the cross product with the edges' unit vectors).
// Margins are relative 4e-12 resp. 1e-12 (the L1 norm bounds r from above), three orders of
// magnitude above the rounding of the products and of the reference's own float64 evaluation.
// Cheap float32 angle guesses for fast_bin (about 1e-5 rad, a few per mille of a bin): whatever they
// get wrong the float64 confirmation rejects, so their accuracy only decides how many points take the
// slow path, never a result.
__device__ __forceinline__ float guess_acosf(float q) {
  if (fabsf(q) > 0.5f) return acosf(q);                       // steep beams: the library routine
  float q2 = q * q;                                           // asin series, error < 3e-6 for |q| <= 0.5
  float p = fmaf(q2, 0.02237216f, 0.03038194f);
  p = fmaf(p, q2, 0.04464286f);
  p = fmaf(p, q2, 0.075f);
  p = fmaf(p, q2, 0.16666667f);
  return 1.57079637f - fmaf(p * q2, q, q);
}
__device__ __forceinline__ float guess_atan2f(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __frcp_rn(mx), t2 = t * t;                   // atan on [0, 1], odd polynomial, error ~1e-5
  float p = fmaf(-0.01172120f, t2, 0.05265332f);
  p = fmaf(p, t2, -0.11643287f);
  p = fmaf(p, t2, 0.19354346f);
  p = fmaf(p, t2, -0.33262347f);
  p = fmaf(p, t2, 0.99997726f);
  float a = p * t;
  a = ay > ax ? 1.57079637f - a : a;
  a = x < 0.f ? 3.14159274f - a : a;
  return y < 0.f ? -a : a;
}

__device__ __forceinline__ bool fast_bin(const Binning &bn, const double *__restrict__ row_cc,
                                         const double *__restrict__ col_dir, float inv_del, float inv_daz,
                                         float elo, float xf, float yf, float zf, double x, double y, double z,
                                         double ss, int &row, int &col) {
  float qf = zf * __frsqrt_rn(xf * xf + yf * yf + zf * zf);
  qf = qf < -1.f ? -1.f : (qf > 1.f ? 1.f : qf);
  int rg = (int)floorf((guess_acosf(qf) - elo) * inv_del);
  int cg = (int)((guess_atan2f(yf, xf) + 3.14159274f) * inv_daz);
  rg = rg < 0 ? 0 : (rg > bn.rows - 1 ? bn.rows - 1 : rg);
  cg = cg < 0 ? 0 : (cg > bn.cols - 1 ? bn.cols - 1 : cg);
  row = rg;
  col = cg;
  const double zz = z * fabs(z);
  const double hi = row_cc[rg == 0 ? 0 : rg + 1], lo = row_cc[rg + 2];
  const double ax = col_dir[2 * cg], ay = col_dir[2 * cg + 1], bx = col_dir[2 * cg + 2], by = col_dir[2 * cg + 3];
  const double mr = 4e-12 * ss, mc = 1e-12 * (fabs(x) + fabs(y) + fabs(z));
  // (non-short-circuit on purpose: six compares and five ANDs instead of five branches)
  return (int)(fabs(zz) < 0.999998 * ss) &                    // acos is ill-conditioned at the poles
         (int)(zz < hi * ss - mr) & (int)(zz > lo * ss + mr) & (int)(ax * y - ay * x > mc) &
         (int)(bx * y - by * x < -mc);
}

__global__ void __launch_bounds__(kPT)
k_project(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int chunks) {
  extern __shared__ __align__(16) double s_tab[];          // [(cols+1)*2] column edges, [rows+2] row edges
  double *s_col = s_tab, *s_row = s_tab + (b.cols + 1) * 2;
  for (int e = threadIdx.x; e < (b.cols + 1) * 2; e += kPT) s_col[e] = w.col_dir[e];
  int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    __syncthreads();                                       // previous scene's row table is no longer read
    for (int e = threadIdx.x; e < b.rows + 2; e += kPT) s_row[e] = w.row_q[(int64_t)s * (b.rows + 2) + e];
    __syncthreads();
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    const bool exact = b.reserved & 1;                       // diagnostic: reference formula only
    const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az);
    const float elo = (float)(bn.min_el + 0.00001);
    uint32_t *queue = w.cand + (int64_t)s * w.cand_stride;  // insert scratch, free during step 0
    int flags = 0;
    // verified float32 guess, 8 points per thread and tile; unconfirmed points are queued for
    // k_project_slow.  A block walks several tiles so that the tables are staged once.
    for (int t0 = blockIdx.x * kTile; t0 < n; t0 += gridDim.x * kTile)
#pragma unroll 2
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      BoxAcc box;
      if (i < n) {
        // float32 points straight from the slab; inserted (float64) points, which only exist when a
        // re-based scene is projected by this kernel, take the queue
        float4 pt = i < n_head ? reinterpret_cast<const float4 *>(b.xyzi)[(int64_t)s * b.cap + i]
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        double x = (double)pt.x, y = (double)pt.y, z = (double)pt.z;
        double ss = x * x + y * y + z * z;
        int row, col;
        if ((int)(!exact) & (int)(i < n_head) &
            (int)fast_bin(bn, s_row, s_col, inv_del, inv_daz, elo, pt.x, pt.y, pt.z, x, y, z, ss, row, col)) {
          int p = row * b.cols + col;
          box.add(row, col);
          if (ss > R3D_EMPTY_DEPTH * R3D_EMPTY_DEPTH) {      // r > 500 (or rounds to it): far list
            int f = atomicAdd(&b.n_far[s], 1);
            if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = p;
            else flags |= R3D_S_FAR_OVERFLOW;
          }
          b.pix[(int64_t)s * b.cap + i] = p;
        } else {
          queue[atomicAdd(&w.n_slow[s], 1)] = (uint32_t)i;
          box.add(0, 0);                                     // unknown pixel: the chunk's box covers
          box.add(b.rows - 1, b.cols - 1);                   // the whole image
        }
      }
      unsigned long long packed = box.wave_pack();
      unsigned long long living = __ballot(i < n);           // every point of the frame is alive at step 0
      int i0 = t0 + k * kPT + (threadIdx.x & ~63);
      if ((threadIdx.x & 63) == 0 && i0 < n) {
        w.chunk_box[(int64_t)s * chunks + (i0 >> 6)] = packed;
        w.alive[(int64_t)s * chunks + (i0 >> 6)] = living;
      }
    }
    flags = wave_or_i32(flags);
    if ((threadIdx.x & 63) == 0 && flags) atomicOr(&b.status[s], flags);
  }
}

// The reference formula (insertion.py:74-76, :104-116) for the points k_project could not confirm:
// none to a handful per scan (points within 1e-12 of a bin edge, or a float32 guess one bin off).
__global__ void __launch_bounds__(kPT)
k_project_slow(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w) {
  int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n_slow = w.n_slow[s], n_head = b.n_head[s];
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    const uint32_t *queue = w.cand + (int64_t)s * w.cand_stride;
    int flags = 0;
    for (int e = blockIdx.x * kPT + threadIdx.x; e < n_slow; e += gridDim.x * kPT) {
      int i = (int)queue[e];
      double x, y, z;
      load_point(b, s, i, n_head, x, y, z);
      BoxAcc unused;
      b.pix[(int64_t)s * b.cap + i] = project_point(b, s, bn, x, y, z, flags, unused);
    }
    if (flags) atomicOr(&b.status[s], flags);
  }
}

// Survivors in original order (insertion.py:472-473 applied once for all steps), float4 + label
// straight into the output arrays.  The 8 alive tests of a thread are issued together, ranks come
// from wave ballots and one small LDS table: a single barrier per tile.
__global__ void __launch_bounds__(kPT)
k_alive_write(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles, int chunks,
              double *rows4, int32_t *n_rows) {
  __shared__ int s_cnt[kPerThread][kPT / 64];           // survivors of (row k, wave)
  __shared__ int s_pre[kPT / 64];
  int cnt = *count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s];
    int t0 = blockIdx.x * kTile;
    if (t0 >= n) continue;
    // survivors of the preceding tiles of this scene (at most a few dozen counts): the tile's offset
    int pre = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += kPT) pre += w.tile_alive[(int64_t)s * tiles + t];
    pre = wave_sum_i32(pre);
    if (lane == 0) s_pre[wave] = pre;
    __syncthreads();
    int tile_base = 0;
#pragma unroll
    for (int v = 0; v < kPT / 64; ++v) tile_base += s_pre[v];
    if (threadIdx.x == 0 && t0 + kTile >= n)                   // the scene's last tile publishes the total
      (rows4 ? n_rows : b.n_out)[s] = tile_base + w.tile_alive[(int64_t)s * tiles + blockIdx.x];
    const int n_head = b.n_head[s];
    const float4 *src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    float4 *dst = reinterpret_cast<float4 *>(b.out_xyzi) + (int64_t)s * b.cap;
    bool flag[kPerThread];
    int rank[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {                  // the alive bits the inserts left
      int i0 = t0 + k * kPT + (threadIdx.x & ~63);
      unsigned long long m = i0 < n ? w.alive[(int64_t)s * chunks + (i0 >> 6)] : 0ull;
      flag[k] = (m >> lane) & 1ull;
      rank[k] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_cnt[k][wave] = __popcll(m);
    }
    __syncthreads();
    int run = tile_base;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      int mine = 0;
#pragma unroll
      for (int v = 0; v < kPT / 64; ++v) {
        if (v == wave) mine = run;
        run += s_cnt[k][v];
      }
      if (flag[k]) {
        int i = t0 + k * kPT + threadIdx.x, o = mine + rank[k];
        if (rows4) {                                        // r3d_batch_export_rows: x y z label as float64
          double x, y, z;
          load_point(b, s, i, n_head, x, y, z);
          double2 *row = reinterpret_cast<double2 *>(rows4 + ((int64_t)s * b.cap + o) * 4);
          row[0] = make_double2(x, y);
          row[1] = make_double2(z, (double)(b.label[(int64_t)s * b.cap + i] & 0xFFFFu));
        } else {
          // (measured and dropped: requesting the thread's 8 points and labels before the first store, 0.32 ms
          // against 0.28 ms; a shifted plain copy for the tiles in which nobody died, 0.30 ms)
          dst[o] = src[i];
          b.out_label[(int64_t)s * b.cap + o] = b.label[(int64_t)s * b.cap + i];
        }
      }
    }
    __syncthreads();                                      // s_cnt is reused by the next scene
  }
}

// check/{f}.bin rows from the log (SS tools/datasets.py:73-75, :86-88; OD :77, :91-93).
__global__ void k_pack_log(r3d_batch_t b, float *__restrict__ check, int check_cols) {
  int s = blockIdx.y;
  int n = b.n_log[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double *l = b.log5 + ((int64_t)s * b.log_cap + i) * 5;
    float *c = check + ((int64_t)s * b.log_cap + i) * check_cols;
    c[0] = (float)l[0];
    c[1] = (float)l[1];
    c[2] = (float)l[2];
    c[3] = (float)l[3];
    if (check_cols == 5) c[4] = (float)l[4];
  }
}

int check_batch(const r3d_batch_t *b) {
  if (!b) return fail(R3D_E_ARG, "batch: null descriptor");
  if (b->B <= 0 || b->rows <= 0 || b->cols <= 0 || b->cap <= 0 || b->log_cap <= 0)
    return fail(R3D_E_ARG, "batch: non-positive shape");
  if (b->cap > (int64_t)1 << 30 || b->rows > 65535 || b->cols > 65535 || (int64_t)b->rows * b->cols > (int64_t)1 << 30)
    return fail(R3D_E_ARG, "batch: cap or range image too large for 32-bit point / pixel ids");
  if (!b->xyzi || !b->label || !b->pix || !b->n_head || !b->n_total || !b->tail_ref || !b->log5 ||
      !b->log_birth || !b->n_log || !b->bounds ||
       !b->far_pix || !b->n_far || !b->rebase || !b->status || !b->out_xyzi ||
      !b->out_label || !b->n_out || !b->workspace)
    return fail(R3D_E_ARG, "batch: null array");
  if (b->cols % 32 != 0)
    return fail(R3D_E_ARG, "batch: cols must be a multiple of 32 (row-aligned bit images)");
  if (((size_t)(b->cols + 1) * 2 + b->rows + 2) * sizeof(double) > 64 * 1024)
    return fail(R3D_E_ARG, "batch: range image too large for the projection kernel's LDS edge tables");
  if (b->workspace_bytes < carve_batch(*b, nullptr).total)
    return fail(R3D_E_WORKSPACE, "batch: workspace smaller than r3d_batch_workspace_bytes()");
  return R3D_OK;
}

// blocks per scene of k_project: each walks ~4 tiles, at least ~2048 blocks in a 256-scene launch
static int project_blocks(const r3d_batch_t &b) {
  int t = tiles_of(b);
  int per = (t + 3) / 4;
  return per < 1 ? 1 : per;
}

static size_t project_lds_bytes(const r3d_batch_t &b) {
  return ((size_t)(b.cols + 1) * 2 + b.rows + 2) * sizeof(double);
}

// bounds -> tables -> project for the scenes of (list, count); rows = block rows of the launches.
static int launch_reproject(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                            const int32_t *count, int rows, hipStream_t st) {
  int tiles = tiles_of(b);
  hipLaunchKernelGGL(k_bounds, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w);
  hipLaunchKernelGGL(k_prepare, dim3(1, rows), dim3(kPT), 0, st, b, list, count, w, tiles);
  hipLaunchKernelGGL(k_project, dim3(project_blocks(b), rows), dim3(kPT), project_lds_bytes(b), st, b, list,
                     count, w, chunks_of(b));
  hipLaunchKernelGGL(k_project_slow, dim3(4, rows), dim3(kPT), 0, st, b, list, count, w);
  R3D_LAUNCHED("reproject kernels");
  return R3D_OK;
}

static int launch_compact(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                          const int32_t *count, int rows, hipStream_t st, double *rows4 = nullptr,
                          int32_t *n_rows = nullptr) {
  int tiles = tiles_of(b);
  hipLaunchKernelGGL(k_alive_write, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, tiles, chunks_of(b),
                     rows4, n_rows);
  R3D_LAUNCHED("compaction kernel");
  return R3D_OK;
}

}  // namespace r3d

using namespace r3d;

extern "C" {

size_t r3d_batch_workspace_bytes(const r3d_batch_t *b) {
  if (!b || b->B <= 0 || b->rows <= 0 || b->cols <= 0 || b->cap <= 0 || b->log_cap <= 0) return 0;
  return carve_batch(*b, nullptr).total;
}

int r3d_batch_create(const r3d_batch_t *b, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  hipLaunchKernelGGL(k_col_table, dim3((b->cols + 1 + 255) / 256), dim3(256), 0, st, *b, w);
  R3D_LAUNCHED("k_col_table");
  return R3D_OK;
}

int r3d_batch_begin(const r3d_batch_t *b, const int32_t *n_points, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!n_points) return fail(R3D_E_ARG, "batch_begin: null n_points");
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  hipLaunchKernelGGL(k_begin_init, dim3((b->B + 255) / 256), dim3(256), 0, st, *b, n_points, w);
  return launch_reproject(*b, w, w.all_list, w.all_count, b->B, st);
}

int r3d_batch_launch_one(const r3d_batch_t *b, int32_t which, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  int tiles = tiles_of(*b);
  switch (which) {
    case R3D_K_BOUNDS:
      hipLaunchKernelGGL(k_bounds, dim3(tiles, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w);
      break;
    case R3D_K_PREPARE:
      hipLaunchKernelGGL(k_prepare, dim3(1, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w, tiles);
      break;
    case R3D_K_PROJECT:
      hipLaunchKernelGGL(k_project, dim3(project_blocks(*b), b->B), dim3(kPT), project_lds_bytes(*b), st, *b,
                         w.all_list, w.all_count, w, chunks_of(*b));
      break;
    case R3D_K_ALIVE_WRITE:
      hipLaunchKernelGGL(k_alive_write, dim3(tiles, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w, tiles,
                         chunks_of(*b), (double *)nullptr, (int32_t *)nullptr);
      break;
    default:
      return fail(R3D_E_ARG, "batch_launch_one: unknown kernel id");
  }
  R3D_LAUNCHED("batch_launch_one");
  return R3D_OK;
}

int r3d_batch_export_rows(const r3d_batch_t *b, double *rows4, int32_t *n_rows, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!rows4 || !n_rows) return fail(R3D_E_ARG, "batch_export_rows: null output");
  BatchWs w = carve_batch(*b, b->workspace);
  return launch_compact(*b, w, w.all_list, w.all_count, b->B, (hipStream_t)stream, rows4, n_rows);
}

int r3d_batch_finish(const r3d_batch_t *b, float *check, int32_t check_cols, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (check && check_cols != 4 && check_cols != 5) return fail(R3D_E_ARG, "batch_finish: check_cols");
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  rc = launch_compact(*b, w, w.all_list, w.all_count, b->B, st);
  if (rc != R3D_OK) return rc;
  if (check) {
    int gx = (int)((b->log_cap + 255) / 256);
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(k_pack_log, dim3(gx, b->B), dim3(256), 0, st, *b, check, (int)check_cols);
    R3D_LAUNCHED("k_pack_log");
  }
  return R3D_OK;
}

}  // extern "C"
