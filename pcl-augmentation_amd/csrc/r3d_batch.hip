// Level 2: B scenes advanced in lock step through K inserts, everything resident in HBM.
//
// The algorithm is the incremental one stated in tests/incremental_model.py and DESIGN.md par.3:
// project every scene once (step 0), then per insert evaluate the sample only on its candidate
// pixels, patch the scene's range image at the visible pixels, stamp them, append the visible
// points; dead points are dropped once, in r3d_batch_finish (or earlier by a "rebase" when the
// elevation bounds may have moved, after which the scene is re-projected like step 0).
//
// Kernels that walk scenes take a (list, count) pair: block row `blockIdx.y` handles scenes
// list[blockIdx.y], list[blockIdx.y + gridDim.y], ... below *count.  With the identity list this
// is "all scenes"; with the rebase list (normally empty) the blocks return at once.
#include "r3d_device.hpp"
#include "r3d_host.hpp"

#include <cstdlib>

namespace r3d {

constexpr int kPT = 256;             // threads of the streaming kernels
constexpr int kPerThread = 8;
constexpr int kTile = kPT * kPerThread;   // points per block tile
constexpr int kST = 1024;            // threads of the per-scene insert kernel
constexpr int kKeyCap = R3D_MAX_SAMPLE;
constexpr int kIdxBits = 13;         // kKeyCap == 1 << kIdxBits
constexpr int kRebaseRows = 16;      // block rows of the (normally idle) rebase launches

struct BatchWs {
  unsigned long long *qkeys;    // [B][2] ordered keys of min / max of z/r
  int32_t *tile_alive;          // [B*tiles]
  uint32_t *cand;               // [B*2*npix] chunk / candidate pixel lists of the insert kernel
  int32_t *all_list;            // [B] identity
  int32_t *all_count;           // [1] = B
  int32_t *rebase_list;         // [B]
  int32_t *n_rebase;            // [1]
  int32_t *rebase_ticket;       // [1]
  unsigned long long *chunk_box; // [B*chunks] rows/cols bounding box of 64 consecutive points
  int32_t *n_proj;              // [B] points covered by the chunk boxes
  double *smp_r;                // [B*R3D_MAX_SAMPLE] range of every sample point (k_insert scratch)
  double *row_q;                // [B*(rows+2)] c*|c|, c = cos of the row edges (entry k: edge k-1)
  double *col_dir;              // [(cols+1)*2] unit vector of every column edge
  double *q_ext;                // [B*2] min and max of z/r (the points that hold the elevation bounds)
  int32_t *n_slow;              // [B] points queued for k_project_slow
  unsigned long long *alive_bits; // [B*chunks] survivor bit of every point (k_alive_count -> k_alive_write)
  int32_t *chain_progress;        // [B] slots of the scene completed by the running k_insert_chain
  int64_t cand_stride;          // uint32 entries of `cand` per scene: max(2*npix, cap)
  size_t total;
};

static int tiles_of(const r3d_batch_t &b) { return (int)((b.cap + kTile - 1) / kTile); }
static int chunks_of(const r3d_batch_t &b) { return (int)((b.cap + 63) / 64); }
static int mask_words(const r3d_batch_t &b) { return (int)(((int64_t)b.rows * b.cols + 31) / 32); }

static BatchWs carve_batch(const r3d_batch_t &b, void *base) {
  BatchWs w;
  Carver c(base);
  int64_t npix = (int64_t)b.rows * b.cols;
  int tiles = tiles_of(b);
  w.qkeys = c.take<unsigned long long>((size_t)b.B * 2);
  w.tile_alive = c.take<int32_t>((size_t)b.B * tiles);
  w.cand_stride = 2 * npix > b.cap ? 2 * npix : b.cap;
  w.cand = c.take<uint32_t>((size_t)b.B * w.cand_stride);
  w.all_list = c.take<int32_t>((size_t)b.B);
  w.all_count = c.take<int32_t>(1);
  w.rebase_list = c.take<int32_t>((size_t)b.B);
  w.n_rebase = c.take<int32_t>(1);
  w.rebase_ticket = c.take<int32_t>(1);
  w.chunk_box = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.n_proj = c.take<int32_t>((size_t)b.B);
  w.smp_r = c.take<double>((size_t)b.B * kKeyCap);
  w.row_q = c.take<double>((size_t)b.B * (b.rows + 2));
  w.col_dir = c.take<double>((size_t)(b.cols + 1) * 2);
  w.q_ext = c.take<double>((size_t)b.B * 2);
  w.n_slow = c.take<int32_t>((size_t)b.B);
  w.alive_bits = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.chain_progress = c.take<int32_t>((size_t)b.B);
  w.total = c.off;
  return w;
}

// ---- cloud access ---------------------------------------------------------------------------
// A cloud point is float32-exact (head, from velodyne .bin) or a float64 inserted point (tail)
// whose exact coordinates live in the log; xyzi holds the float32 rounding of tail points so the
// output .bin bytes are a plain copy.
__device__ __forceinline__ void load_point(const r3d_batch_t &b, int s, int i, int n_head, double &x,
                                           double &y, double &z) {
  if (i < n_head) {
    float4 p = reinterpret_cast<const float4 *>(b.xyzi)[(int64_t)s * b.cap + i];
    x = (double)p.x;
    y = (double)p.y;
    z = (double)p.z;
  } else {
    int lr = b.tail_ref[(int64_t)s * b.log_cap + (i - n_head)];
    const double *q = b.log5 + ((int64_t)s * b.log_cap + lr) * 5;
    x = q[0];
    y = q[1];
    z = q[2];
  }
}

__device__ __forceinline__ bool point_alive_at(const r3d_batch_t &b, int s, int i, int p, int n_head,
                                               int npix, int words) {
  if (p < 0) return false;                              // entombed by an earlier rebase
  if (i < n_head) return !((b.ever[(int64_t)s * words + (p >> 5)] >> (p & 31)) & 1u);
  int lr = b.tail_ref[(int64_t)s * b.log_cap + (i - n_head)];
  return (int)b.stamp[(int64_t)s * npix + p] <= b.log_birth[(int64_t)s * b.log_cap + lr];
}

__device__ __forceinline__ bool point_alive(const r3d_batch_t &b, int s, int i, int n_head, int npix,
                                            int words) {
  int p = b.pix[(int64_t)s * b.cap + i];
  if (p < 0) return false;                              // entombed by an earlier rebase
  if (i < n_head) return !((b.ever[(int64_t)s * words + (p >> 5)] >> (p & 31)) & 1u);
  int lr = b.tail_ref[(int64_t)s * b.log_cap + (i - n_head)];
  return (int)b.stamp[(int64_t)s * npix + p] <= b.log_birth[(int64_t)s * b.log_cap + lr];
}

// ---- step 0 / rebase: bounds ------------------------------------------------------------------
__global__ void k_begin_init(r3d_batch_t b, const int32_t *n_points, BatchWs w) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {
    *w.all_count = b.B;
    *w.n_rebase = 0;
    *w.rebase_ticket = 0;
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
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      if (i < n) {
        double x, y, z;
        load_point(b, s, i, n_head, x, y, z);
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

// After k_bounds: the elevation bounds (insertion.py:78-79) = acos of the extreme z/r, the row-edge
// table of the verified fast projection (k_project), and the reset of the visibility stamps, in one
// launch.  Only pixels whose `ever` bit is set carry a stamp, so the reset walks the bit image
// (20 KB per scene) instead of the stamp image (322 KB per scene).
//
// Tables of the fast projection: a bin guessed in float32 is accepted only if the point lies
// strictly inside that bin's edges, tested in float64 on monotone images of the edges -- cos of
// the row edges against z/r, and the sign of the cross product with the unit vector of the column
// edges -- with a margin far above the rounding of either side.
__global__ void __launch_bounds__(kPT)
k_prepare(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w) {
  __shared__ double s_b[2];
  __shared__ int s_ok;
  int cnt = *count;
  int64_t npix = (int64_t)b.rows * b.cols;
  int words = (int)((npix + 31) / 32);
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0) {
        unsigned long long kmin = w.qkeys[2 * s + 0], kmax = w.qkeys[2 * s + 1];
        s_ok = kmin != ~0ull;
        if (!s_ok) {                          // no valid point: the reference raises (insertion.py:78)
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
      __syncthreads();
    }
    uint16_t *st = b.stamp + (int64_t)s * npix;
    uint32_t *ev = b.ever + (int64_t)s * words;
    for (int p = blockIdx.x * kPT + threadIdx.x; p < words; p += gridDim.x * kPT) {
      uint32_t e = ev[p];
      if (!e) continue;
      ev[p] = 0u;
      while (e) {
        int bit = __ffs(e) - 1;
        e &= e - 1;
        if ((int64_t)p * 32 + bit < npix) st[(int64_t)p * 32 + bit] = 0;
      }
    }
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
__device__ __forceinline__ unsigned long long pack_box(int rmin, int rmax, int cmin, int cmax) {
  return (unsigned long long)(rmin & 0xFFFF) | ((unsigned long long)(rmax & 0xFFFF) << 16) |
         ((unsigned long long)(cmin & 0xFFFF) << 32) | ((unsigned long long)(cmax & 0xFFFF) << 48);
}

// Projects point i of scene s (if valid) and returns its pixel; accumulates the wave's box.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct BoxAcc {
  u16x2 lo = {0xFFFF, 0xFFFF}, hi = {0, 0};        // (row, col) minima and maxima
  __device__ __forceinline__ void add(int row, int col) {
    u16x2 v = {(unsigned short)row, (unsigned short)col};
    lo = __builtin_elementwise_min(lo, v);
    hi = __builtin_elementwise_max(hi, v);
  }
  __device__ __forceinline__ unsigned long long wave_pack() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                 // two packed 16-bit reductions per step
      int tl = __shfl_xor(__builtin_bit_cast(int, lo), o, 64);
      int th = __shfl_xor(__builtin_bit_cast(int, hi), o, 64);
      lo = __builtin_elementwise_min(lo, __builtin_bit_cast(u16x2, tl));
      hi = __builtin_elementwise_max(hi, __builtin_bit_cast(u16x2, th));
    }
    return pack_box(lo.x, hi.x, lo.y, hi.y);           // rmin > rmax: empty box
  }
};

__device__ __forceinline__ int project_point(const r3d_batch_t &b, int s, const Binning &bn, double x,
                                             double y, double z, int &flags, BoxAcc &box) {
  Sph sp = spherical(x, y, z);
  int row, col, p = 0;
  int ok = bin_point(bn, sp.az, sp.el, row, col);
  if (!(ok & 1)) flags |= isfinite(sp.el) ? R3D_S_ROW_RANGE : R3D_S_NONFINITE;   // assert :110
  else if (!(ok & 2)) flags |= R3D_S_COL_RANGE;                                   // assert :112
  else {
    p = row * b.cols + col;
    box.add(row, col);
    if (sp.r > R3D_EMPTY_DEPTH) {           // "first hit overwrites the 500": insertion.py:122-125
      int f = atomicAdd(&b.n_far[s], 1);
      if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = p;
      else flags |= R3D_S_FAR_OVERFLOW;
    }
  }
  return p;
}

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
    if (blockIdx.x == 0 && threadIdx.x == 0) w.n_proj[s] = n;
    __syncthreads();                                       // previous scene's row table is no longer read
    for (int e = threadIdx.x; e < b.rows + 2; e += kPT) s_row[e] = w.row_q[(int64_t)s * (b.rows + 2) + e];
    __syncthreads();
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    const bool exact = b.reserved & 1;                       // diagnostic: reference formula only
    const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az);
    const float elo = (float)(bn.min_el + 0.00001);
    uint32_t *queue = w.cand + (int64_t)s * w.cand_stride;  // k_insert scratch, free during step 0
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
      int i0 = t0 + k * kPT + (threadIdx.x & ~63);
      if ((threadIdx.x & 63) == 0 && i0 < n) w.chunk_box[(int64_t)s * chunks + (i0 >> 6)] = packed;
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

// ---- one insert candidate per scene ------------------------------------------------------------
// One workgroup evaluates one placement candidate against one scene (insertion.py:455-526).
// Everything it needs lives in a WINDOW of the range image around the sample: rows
// [r_lo, r_hi] x one or two column intervals of whole 32-pixel words (two when the object
// straddles the azimuth seam).  Inside the window:
//   * bit images (sample / scene occupancy, their closings, candidates, visible pixels) are
//     window-local words in LDS; the 5-row x 3-column closing of closing.py:20 is word-parallel
//     shifts with ORs / ANDs, windows clipped at the image border like the reference's;
//   * the scene's depths are min-reduced from the alive points into an LDS tile (or, when the
//     window is too large for LDS, into the global scratch image `grid`);
//   * the sample's depths live in an LDS array indexed by the rank of the pixel among the
//     sample's occupied pixels (or in the global scratch image `sgrid`).
// Sorts keys[0, pw) ascending; pw is a power of two, pw <= E * blockDim.x, pw >= 64.
template <int E>
__device__ __forceinline__ void block_bitonic_sort(uint32_t *keys, int pw, int tid) {
  const int active = pw / E;                  // threads that own elements
  uint32_t v[E];
  const bool own = tid < active;
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = own ? keys[tid * E + e] : 0xFFFFFFFFu;
  for (int k = 2; k <= pw; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64 * E) {                       // partner lives in another wave: exchange through LDS
        __syncthreads();
        if (own) {
#pragma unroll
          for (int e = 0; e < E; ++e) keys[tid * E + e] = v[e];
        }
        __syncthreads();
        if (own) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            int i = tid * E + e;
            uint32_t o = keys[i ^ j];
            bool keep_min = ((i & j) == 0) == ((i & k) == 0);
            v[e] = keep_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
          }
        }
      } else if (j >= E) {                     // partner is another lane of this wave
        const int lane_mask = j / E;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          int i = tid * E + e;
          uint32_t o = (uint32_t)__shfl_xor((int)v[e], lane_mask, 64);
          bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          v[e] = keep_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
        }
      } else {                                 // partner is another register of this thread
#pragma unroll
        for (int e = 0; e < E; ++e) {
          int e2 = e ^ j;
          if (e2 > e) {
            int i = tid * E + e;
            bool up = (i & k) == 0;
            uint32_t a = v[e], c = v[e2];
            bool swap = (a > c) == up;
            v[e] = swap ? c : a;
            v[e2] = swap ? a : c;
          }
        }
      }
    }
  }
  __syncthreads();
  if (own) {
#pragma unroll
    for (int e = 0; e < E; ++e) keys[tid * E + e] = v[e];
  }
  __syncthreads();
}

struct Window {
  int r_lo, r_hi, n_iv, jl[2], jh[2], nj0, njw, nrw, cols;
  // window-local word index of image word (row r, word j), -1 outside the window
  __device__ __forceinline__ int lword(int r, int j) const {
    if (r < r_lo || r > r_hi) return -1;
    int k;
    if (j >= jl[0] && j <= jh[0]) k = j - jl[0];
    else if (n_iv > 1 && j >= jl[1] && j <= jh[1]) k = nj0 + j - jl[1];
    else return -1;
    return (r - r_lo) * njw + k;
  }
  __device__ __forceinline__ int row_of(int e) const { return r_lo + e / njw; }      // e: local word
  __device__ __forceinline__ int word_of(int e) const {
    int k = e % njw;
    return k < nj0 ? jl[0] + k : jl[1] + (k - nj0);
  }
  __device__ __forceinline__ int lpix_rc(int r, int c) const {  // window-local pixel, -1 outside
    int lw = lword(r, c >> 5);
    return lw < 0 ? -1 : (lw << 5) + (c & 31);
  }
  __device__ __forceinline__ int lpix(int q) const {
    int r = q / cols;
    return lpix_rc(r, q - r * cols);
  }
};

struct WinImage {
  uint32_t *w;
  __device__ __forceinline__ bool get_local(int lp) const { return (w[lp >> 5] >> (lp & 31)) & 1u; }
  __device__ __forceinline__ void set_local(int lp) { atomicOr(&w[lp >> 5], 1u << (lp & 31)); }
  __device__ __forceinline__ uint32_t word(const Window &win, int r, int j) const {
    int lw = win.lword(r, j);
    return lw < 0 ? 0u : w[lw];
  }
};

// OR of a word with its horizontal neighbours' bits (columns c-1, c, c+1), clipped at the row ends.
__device__ __forceinline__ uint32_t hor3(const WinImage &m, const Window &win, int r, int j) {
  uint32_t c = m.word(win, r, j), l = m.word(win, r, j - 1), rr = m.word(win, r, j + 1);
  return c | (c << 1) | (l >> 31) | (c >> 1) | (rr << 31);
}
// AND of the same three columns; a neighbour outside the IMAGE does not constrain (erosion border).
__device__ __forceinline__ uint32_t hand3(const WinImage &m, const Window &win, int r, int j, int wpr) {
  uint32_t c = m.word(win, r, j);
  uint32_t l = j > 0 ? (m.word(win, r, j - 1) >> 31) : 1u;
  uint32_t rr = j < wpr - 1 ? (m.word(win, r, j + 1) << 31) : 0x80000000u;
  return c & ((c << 1) | l) & ((c >> 1) | rr);
}

// closing.py:44-57 on up to 15 already loaded keys (R3D_SENT = empty): sum over the occupied ones,
// drow outer / dcolumn inner, divided by their count.
__device__ __forceinline__ double mean_of_keys(const unsigned long long (&v)[15]) {
  double sum = 0.0;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 15; ++k)
    if (v[k] != R3D_SENT) {
      ++cnt;
      sum += key_depth(v[k]);
    }
  return cnt ? sum / (double)cnt : R3D_EMPTY_DEPTH;
}

// Diagnostic builds (make STAMPS=1) record a 100 MHz wall-clock stamp per phase in the first bytes
// of the scene's out_xyzi slab (scratch until r3d_batch_finish); tools/stamps_insert.py reads them.
#ifdef R3D_STAMPS
#define STAMP(i)                                                                                   \
  do {                                                                                             \
    __syncthreads();                                                                               \
    if (tid == 0) reinterpret_cast<long long *>(b.out_xyzi + (int64_t)s * b.cap * 4)[i] = wall_clock64(); \
  } while (0)
#else
#define STAMP(i)
#endif
constexpr int kLdsBytes = 160 * 1024;
constexpr int kLdsFixed = 256 + kKeyCap / 8;      // counters + out-of-bounds bits

// One placement candidate of scene s (one workgroup).  CHAIN: the caller re-bases the scene itself
// when the return value says so, instead of the list that k_rebase reads.
template <bool CHAIN>
__device__ __forceinline__ bool
insert_scene(const r3d_batch_t &b, const double *__restrict__ samples5, const int64_t *__restrict__ sample_off,
             const int32_t *__restrict__ min_points, const int32_t *__restrict__ active, int step,
             int32_t *__restrict__ n_visible, int32_t *__restrict__ accepted, const BatchWs &w, int chunks,
             const int s, unsigned char *smem) {
  const int tid = threadIdx.x;
  const int rows = b.rows, cols = b.cols;
  const int npix = rows * cols;
  const int words = npix >> 5, wpr = cols >> 5;
  int *s_misc = reinterpret_cast<int *>(smem);
  int *s_nvalid = s_misc + 0, *s_ncand = s_misc + 1, *s_rebase = s_misc + 2, *s_flags = s_misc + 3;
  int *s_rmin = s_misc + 4, *s_rmax = s_misc + 5;                 // sample row range
  int *s_cmin = s_misc + 6, *s_cmax = s_misc + 8;                 // [2] column range per image half
  int *s_nlist = s_misc + 10, *s_carry = s_misc + 11;
  int *s_ext = s_misc + 12;                                       // [2] window pixel of a max / min elevation point
  int *s_scan = s_misc + 14;                                      // [kST/64 + 1]
  uint32_t *s_oob = reinterpret_cast<uint32_t *>(smem + 256);     // [kKeyCap/32] el outside bounds
  uint32_t *s_keys = reinterpret_cast<uint32_t *>(smem + kLdsFixed);   // [pw] sorted (pixel, index)

  const int64_t off = sample_off[s];
  const int64_t m64 = sample_off[s + 1] - off;
  const bool on = (!active || active[s]) && m64 > 0 && m64 <= kKeyCap;
  if (!on) {
    if (tid == 0) {
      n_visible[s] = 0;
      accepted[s] = 0;
      if (m64 > kKeyCap && (!active || active[s])) atomicOr(&b.status[s], R3D_S_SAMPLE_TOO_LARGE);
    }
    return false;
  }
  const int m = (int)m64;
  int pw = 64;
  while (pw < m) pw <<= 1;

  for (int i = tid; i < kKeyCap / 32; i += kST) s_oob[i] = 0u;
  if (tid < 14) s_misc[tid] = (tid == 4 || tid == 6 || tid == 7) ? 0x7FFFFFFF : (tid == 5 || tid == 8 || tid == 9 || tid >= 12) ? -1 : 0;
  __syncthreads();

  const Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], rows, cols);
  unsigned long long *grid = (unsigned long long *)b.grid + (int64_t)s * npix;
  unsigned long long *sgrid = (unsigned long long *)b.sgrid + (int64_t)s * npix;
  double *smp_r = w.smp_r + (int64_t)s * kKeyCap;
  const double *rows5 = samples5 + off * 5;

  STAMP(0);
  // -- 1. project the sample with the scene's bounds, sample=True (insertion.py:455-459) ---------
  {
    // column ranges are kept per image half so that an object across the azimuth seam (columns
    // 0 and cols-1) yields two narrow windows instead of one full-width window
    const int half = cols >> 1;
    int rmin = 0x7FFFFFFF, rmax = -1, cmin0 = 0x7FFFFFFF, cmax0 = -1, cmin1 = 0x7FFFFFFF, cmax1 = -1;
    int nval = 0, flags = 0;
    for (int j = tid; j < pw; j += kST) {
      uint32_t key = 0xFFFFFFFFu;
      if (j < m) {
        const double *q = rows5 + (int64_t)j * 5;
        Sph sp = spherical(q[0], q[1], q[2]);
        int row, col;
        int ok = bin_point(bn, sp.az, sp.el, row, col);
        if (!isfinite(sp.el) || !isfinite(sp.az)) {
          flags |= R3D_S_NONFINITE;
        } else if (ok & 1) {                         // rows outside [0, rows) are skipped (:107-108)
          if (!(ok & 2)) {
            flags |= R3D_S_COL_RANGE;                // assert :112
          } else {
            key = ((uint32_t)row << 16) | (uint32_t)col;    // re-keyed by window pixel below
            smp_r[j] = sp.r;
            ++nval;
            rmin = row < rmin ? row : rmin;
            rmax = row > rmax ? row : rmax;
            if (col < half) {
              cmin0 = col < cmin0 ? col : cmin0;
              cmax0 = col > cmax0 ? col : cmax0;
            } else {
              cmin1 = col < cmin1 ? col : cmin1;
              cmax1 = col > cmax1 ? col : cmax1;
            }
            if (sp.el < bn.min_el || sp.el > bn.max_el) atomicOr(&s_oob[j >> 5], 1u << (j & 31));
          }
        }
      }
      s_keys[j] = key;
    }
    nval = wave_sum_i32(nval);
    flags = wave_or_i32(flags);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      int t;
      t = __shfl_xor(rmin, o, 64); rmin = t < rmin ? t : rmin;
      t = __shfl_xor(rmax, o, 64); rmax = t > rmax ? t : rmax;
      t = __shfl_xor(cmin0, o, 64); cmin0 = t < cmin0 ? t : cmin0;
      t = __shfl_xor(cmax0, o, 64); cmax0 = t > cmax0 ? t : cmax0;
      t = __shfl_xor(cmin1, o, 64); cmin1 = t < cmin1 ? t : cmin1;
      t = __shfl_xor(cmax1, o, 64); cmax1 = t > cmax1 ? t : cmax1;
    }
    if ((tid & 63) == 0) {
      atomicAdd(s_nvalid, nval);
      if (flags) atomicOr(s_flags, flags);
      atomicMin(s_rmin, rmin);
      atomicMax(s_rmax, rmax);
      atomicMin(s_cmin + 0, cmin0);
      atomicMax(s_cmax + 0, cmax0);
      atomicMin(s_cmin + 1, cmin1);
      atomicMax(s_cmax + 1, cmax1);
    }
  }
  __syncthreads();

  const int nvalid = *s_nvalid;
  const int n_far = b.n_far[s] < R3D_FAR_CAP ? b.n_far[s] : R3D_FAR_CAP;
  const int n_total = b.n_total[s], n_head = b.n_head[s], n_log = b.n_log[s];

  STAMP(1);
  // -- 3. the window: candidates lie within 2 rows / 1 column of a sample pixel, their closing
  // looks 4 rows / 2 columns further.  Pixels deeper than 500 m (far list) can be visible
  // anywhere: whole image.
  Window win;
  win.cols = cols;
  win.r_lo = 0;
  win.r_hi = rows - 1;
  win.n_iv = 1;
  win.jl[0] = 0;
  win.jh[0] = wpr - 1;
  win.jl[1] = win.jh[1] = 0;
  if (n_far == 0 && nvalid == 0) {
    win.r_hi = -1;                                       // nothing can be visible: empty window
  } else if (n_far == 0) {
    win.r_lo = *s_rmin - 6 < 0 ? 0 : *s_rmin - 6;
    win.r_hi = *s_rmax + 6 > rows - 1 ? rows - 1 : *s_rmax + 6;
    win.n_iv = 0;
    for (int h = 0; h < 2; ++h) {
      if (s_cmax[h] < 0) continue;
      int lo = s_cmin[h] - 3 < 0 ? 0 : s_cmin[h] - 3;
      int hi = s_cmax[h] + 3 > cols - 1 ? cols - 1 : s_cmax[h] + 3;
      int a = lo >> 5, z = hi >> 5;
      if (win.n_iv == 1 && a <= win.jh[0] + 1) win.jh[0] = z > win.jh[0] ? z : win.jh[0];   // merge
      else {
        win.jl[win.n_iv] = a;
        win.jh[win.n_iv] = z;
        ++win.n_iv;
      }
    }
  }
  win.nj0 = win.jh[0] - win.jl[0] + 1;
  win.njw = win.nj0 + (win.n_iv > 1 ? win.jh[1] - win.jl[1] + 1 : 0);
  win.nrw = win.r_hi - win.r_lo + 1;
  const int ww = win.nrw * win.njw;                       // window words
  const int wpx = ww << 5;                                // window pixels

  // LDS carve: keys | 5 bit images + rank | sample depths | scene depth tile
  int carve = (kLdsFixed + pw * 4 + 7) & ~7;
  uint32_t *s_img = reinterpret_cast<uint32_t *>(smem + carve);
  carve = (carve + 6 * ww * 4 + 7) & ~7;
  WinImage A{s_img};                 // sample occupancy
  WinImage T{s_img + ww};            // scratch: dilations, candidate mask, then visible pixels
  WinImage Cs{s_img + 2 * ww};       // sample closed
  WinImage D{s_img + 3 * ww};        // scene occupancy
  WinImage E{s_img + 4 * ww};        // scene closed
  uint32_t *s_rank = s_img + 5 * ww; // occupied sample pixels before each window word
  const bool s_lds = carve + nvalid * 8 <= kLdsBytes;
  unsigned long long *s_sdepth = reinterpret_cast<unsigned long long *>(smem + carve);
  if (s_lds) carve += nvalid * 8;
  const bool c_lds = carve + (int64_t)wpx * 8 <= kLdsBytes;
  unsigned long long *s_ctile = reinterpret_cast<unsigned long long *>(smem + carve);

  if (carve > kLdsBytes) {
    // the bit images of the window do not fit next to the keys (only possible for the whole-image
    // window of a far-pixel list on a range image much larger than the reference's): give up on
    // this candidate, loudly
    if (tid == 0) {
      atomicOr(&b.status[s], R3D_S_WINDOW_TOO_LARGE);
      n_visible[s] = 0;
      accepted[s] = 0;
    }
    return false;
  }
  for (int i = tid; i < 6 * ww; i += kST) s_img[i] = 0u;
  if (s_lds)
    for (int i = tid; i < nvalid; i += kST) s_sdepth[i] = R3D_SENT;
  if (c_lds)
    for (int i = tid; i < wpx; i += kST) s_ctile[i] = R3D_SENT;
  // keys = (window pixel, sample index): every valid sample pixel lies inside the window
  for (int j = tid; j < pw; j += kST) {
    uint32_t rc = s_keys[j];
    if (rc != 0xFFFFFFFFu)
      s_keys[j] = ((uint32_t)win.lpix_rc((int)(rc >> 16), (int)(rc & 0xFFFF)) << kIdxBits) | (uint32_t)j;
  }
  __syncthreads();

  STAMP(2);
  // -- 2. sort (pixel, sample index): the order of visible_sample (insertion.py:474-482) ---------
  // Bitonic network; thread t owns elements t*E .. t*E+E-1.  Strides below E swap registers,
  // strides below 64*E are wave shuffles, only the strides that cross waves go through LDS.
  switch (pw <= kST ? 1 : pw / kST) {
    case 1: block_bitonic_sort<1>(s_keys, pw, tid); break;
    case 2: block_bitonic_sort<2>(s_keys, pw, tid); break;
    case 4: block_bitonic_sort<4>(s_keys, pw, tid); break;
    default: block_bitonic_sort<8>(s_keys, pw, tid); break;
  }

  auto global_pix = [&](int lp) {
    return win.row_of(lp >> 5) * cols + (win.word_of(lp >> 5) << 5) + (lp & 31);
  };

  STAMP(3);
  // -- 3a. sample occupancy, rank of every occupied sample pixel -----------------------------------
  for (int k = tid; k < nvalid; k += kST) {
    int lp = (int)(s_keys[k] >> kIdxBits);
    if (k == 0 || (int)(s_keys[k - 1] >> kIdxBits) != lp) A.set_local(lp);
  }
  __syncthreads();
  for (int base = 0; base < ww; base += kST) {           // exclusive prefix popcount over the words
    int e = base + tid;
    int c = e < ww ? __popc(A.w[e]) : 0;
    int tot;
    int ex = block_escan_i32(c, s_scan, tot);
    int carry0 = *s_carry;
    if (e < ww) s_rank[e] = (uint32_t)(carry0 + ex);
    __syncthreads();
    if (tid == 0) *s_carry = carry0 + tot;
    __syncthreads();
  }
  auto sample_rank = [&](int lp) {
    return (int)s_rank[lp >> 5] + __popc(A.w[lp >> 5] & ((1u << (lp & 31)) - 1u));
  };
  // sample depth per occupied pixel = min r over its points (insertion.py:118-125); runs are short
  for (int k = tid; k < nvalid; k += kST) {
    uint32_t key = s_keys[k];
    int lp = (int)(key >> kIdxBits);
    if (k != 0 && (int)(s_keys[k - 1] >> kIdxBits) == lp) continue;
    double best = smp_r[key & (kKeyCap - 1)];
    for (int k2 = k + 1; k2 < nvalid && (int)(s_keys[k2] >> kIdxBits) == lp; ++k2) {
      double r2 = smp_r[s_keys[k2] & (kKeyCap - 1)];
      best = r2 < best ? r2 : best;
    }
    if (s_lds) s_sdepth[sample_rank(lp)] = depth_key(best);
    else sgrid[global_pix(lp)] = depth_key(best);
  }
  auto sample_key = [&](int q, int lp) -> unsigned long long {   // lp = window-local pixel of q
    if (!A.get_local(lp)) return R3D_SENT;
    return s_lds ? s_sdepth[sample_rank(lp)]
                 : __hip_atomic_load(&sgrid[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  STAMP(4);
  // -- 3b. the scene's range image inside the window, built from the points (insertion.py:118-125)
  // Alive points whose pixel lies in the window min-reduce into the tile: first the 64-point
  // chunks whose bounding box touches the window, then the points appended since the last
  // projection.  Dead points (their pixel was visible at a later step) are skipped, which is what
  // culling them (:472-473) does to the image.
  const double q_min = w.q_ext[2 * s + 0], q_max = w.q_ext[2 * s + 1];
  uint32_t *cand = w.cand + (int64_t)s * w.cand_stride;  // [npix] window-local pixel, then [npix] (row, col)
  uint32_t *cand_rc = cand + npix;
  // 4 points per thread are taken through the dependent loads (pixel -> alive -> coordinates)
  // stage by stage, so that the loads of one stage are in flight together.
  auto reduce_points = [&](const int (&idx)[4]) {
    int p[4], lp[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = idx[u] >= 0 ? b.pix[(int64_t)s * b.cap + idx[u]] : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) lp[u] = (idx[u] >= 0 && p[u] >= 0) ? win.lpix(p[u]) : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) ok[u] = lp[u] >= 0 && point_alive_at(b, s, idx[u], p[u], n_head, npix, words);
    double x[4], y[4], z[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = 1.0;
      y[u] = z[u] = 0.0;
      if (ok[u]) load_point(b, s, idx[u], n_head, x[u], y[u], z[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      double r = sqrt(x[u] * x[u] + y[u] * y[u] + z[u] * z[u]);
      unsigned long long key = depth_key(r);
      if (c_lds) atomicMin(&s_ctile[lp[u]], key);
      else atomicMin(&grid[p[u]], key);
      // the points that hold the elevation bounds (max elevation = acos(min z/r)): if the pixel of
      // one of them turns out visible it is culled and the bounds may move (any holder will do)
      double q = z[u] / r;
      if (q == q_min) s_ext[0] = lp[u];
      if (q == q_max) s_ext[1] = lp[u];
    }
  };
  if (ww > 0) {
    const int n_proj = w.n_proj[s] < n_total ? w.n_proj[s] : n_total;
    const int n_chunks = (n_proj + 63) >> 6;
    const unsigned long long *boxes = w.chunk_box + (int64_t)s * chunks;
    for (int c = tid; c < n_chunks; c += kST) {
      unsigned long long bx = boxes[c];
      int rmin = (int)(bx & 0xFFFF), rmax = (int)((bx >> 16) & 0xFFFF);
      int jmin = (int)((bx >> 32) & 0xFFFF) >> 5, jmax = (int)((bx >> 48) & 0xFFFF) >> 5;
      bool hit = rmin <= win.r_hi && rmax >= win.r_lo &&
                 ((jmin <= win.jh[0] && jmax >= win.jl[0]) ||
                  (win.n_iv > 1 && jmin <= win.jh[1] && jmax >= win.jl[1]));
      if (hit) cand[atomicAdd(s_nlist, 1)] = (uint32_t)c;
    }
    __syncthreads();
    const int npts = *s_nlist * 64;
    for (int e0 = tid; e0 < npts; e0 += 4 * kST) {
      int idx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int e = e0 + u * kST;
        int i = e < npts ? (int)(cand[e >> 6] << 6) + (e & 63) : -1;
        idx[u] = i < n_proj ? i : -1;
      }
      reduce_points(idx);
    }
    for (int i0 = n_proj + tid; i0 < n_total; i0 += 4 * kST) {
      int idx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) idx[u] = i0 + u * kST < n_total ? i0 + u * kST : -1;
      reduce_points(idx);
    }
  }
  __syncthreads();     // LDS tile complete / every global atomic performed (vmcnt(0) at the barrier)
  auto scene_key = [&](int q, int lp) -> unsigned long long {
    return c_lds ? s_ctile[lp] : __hip_atomic_load(&grid[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  STAMP(5);
  // scene occupancy bits: one window word per thread
  for (int e = tid; e < ww; e += kST) {
    int q0 = win.row_of(e) * cols + (win.word_of(e) << 5);
    uint32_t bits = 0;
    for (int bit = 0; bit < 32; ++bit) bits |= (scene_key(q0 + bit, (e << 5) + bit) != R3D_SENT ? 1u : 0u) << bit;
    D.w[e] = bits;
  }
  __syncthreads();

  STAMP(6);
  // -- 4. closing of both occupancies (closing.py:9-23) by word-parallel dilate / erode ----------
  // Exact on every row at least 2 inside the window (or at the image border): candidates are.
  for (int pass = 0; pass < 4; ++pass) {
    const WinImage &src = pass == 0 ? A : pass == 2 ? D : T;
    WinImage &dst = pass == 0 ? T : pass == 1 ? Cs : pass == 2 ? T : E;
    const bool erode = pass & 1;
    for (int e = tid; e < ww; e += kST) {
      int r = win.row_of(e), j = win.word_of(e);
      uint32_t acc = erode ? 0xFFFFFFFFu : 0u;
      for (int dr = -2; dr <= 2; ++dr) {
        int rr = r + dr;
        if (rr < 0 || rr >= rows) continue;
        if (erode) acc &= hand3(src, win, rr, j, wpr);
        else acc |= hor3(src, win, rr, j);
      }
      dst.w[e] = acc;
    }
    __syncthreads();
  }

  STAMP(7);
  // -- 5. candidate pixels: where the sample is closed, plus the far neighbourhoods --------------
  for (int e = tid; e < ww; e += kST) T.w[e] = Cs.w[e];
  __syncthreads();
  for (int f = tid; f < n_far; f += kST) {
    int p = b.far_pix[(int64_t)s * R3D_FAR_CAP + f];
    int r = p / cols, c = p - r * cols;
    for (int dr = -2; dr <= 2; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        if (rr >= 0 && rr < rows && cc >= 0 && cc < cols) T.set_local(win.lpix(rr * cols + cc));
      }
  }
  __syncthreads();
  for (int e = tid; e < ww; e += kST) {
    uint32_t bits = T.w[e];
    if (!bits) continue;
    int pos = atomicAdd(s_ncand, __popc(bits));
    const uint32_t rc0 = ((uint32_t)win.row_of(e) << 16) | (uint32_t)(win.word_of(e) << 5);
    while (bits) {
      int bit = __ffs(bits) - 1;
      bits &= bits - 1;
      cand[pos] = (uint32_t)((e << 5) + bit);            // window-local pixel
      cand_rc[pos++] = rc0 + (uint32_t)bit;              // (row << 16) | column
    }
  }
  __syncthreads();
  const int ncand = *s_ncand;
  for (int e = tid; e < ww; e += kST) T.w[e] = 0u;
  __syncthreads();
  WinImage &vis = T;

  STAMP(8);
  // -- 6. visibility on the candidates: smoothed sample depth < smoothed scene depth (:461-467) --
  for (int ci = tid; ci < ncand; ci += kST) {
    int lp = (int)cand[ci];
    int r = (int)(cand_rc[ci] >> 16), c = (int)(cand_rc[ci] & 0xFFFF);
    int q = r * cols + c;
    double sd = R3D_EMPTY_DEPTH, cd = R3D_EMPTY_DEPTH;
    bool s_hole = !A.get_local(lp) && Cs.get_local(lp);
    bool c_hole = !D.get_local(lp) && E.get_local(lp);
    if (A.get_local(lp)) sd = key_depth(sample_key(q, lp));
    if (D.get_local(lp)) cd = key_depth(scene_key(q, lp));
    // hole means (closing.py:44-57): the 15 neighbour keys are gathered first, one image at a time
    // (one register array), then summed in the reference's order
    auto gather15 = [&](bool scene, unsigned long long (&v)[15]) {
#pragma unroll
      for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
          int rr = r + dr, cc = c + dc, k = (dr + 2) * 3 + (dc + 1);
          v[k] = R3D_SENT;
          if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
          int q2 = rr * cols + cc, lp2 = win.lpix_rc(rr, cc);   // inside the window: holes are >= 2 rows in
          if (lp2 < 0) continue;
          v[k] = scene ? scene_key(q2, lp2) : sample_key(q2, lp2);
        }
    };
    if (s_hole) {
      unsigned long long v[15];
      gather15(false, v);
      sd = mean_of_keys(v);
    }
    if (c_hole) {
      unsigned long long v[15];
      gather15(true, v);
      cd = mean_of_keys(v);
    }
    if (sd < cd) vis.set_local(lp);
  }
  __syncthreads();

  STAMP(9);
  // -- 7. count the visible sample points, accept test (insertion.py:511-517) --------------------
  int mine = 0;
  for (int k = tid; k < nvalid; k += kST) mine += vis.get_local((int)(s_keys[k] >> kIdxBits)) ? 1 : 0;
  int nvis;
  (void)block_escan_i32(mine, s_scan, nvis);
  int need = min_points[s];
  bool accept = nvis > 0 && nvis >= need;
  if (accept && ((int64_t)n_total + nvis > b.cap || (int64_t)n_log + nvis > b.log_cap)) {
    accept = false;
    if (tid == 0) atomicOr(&b.status[s], R3D_S_CAPACITY);
  }

  STAMP(10);
  // -- 8. commit: append (insertion.py:526), stamp the visible pixels ------------------------------
  if (accept) {
    int base = 0;
    for (int k0 = 0; k0 < nvalid; k0 += kST) {
      int k = k0 + tid;
      uint32_t key = k < nvalid ? s_keys[k] : 0u;
      int lp = (int)(key >> kIdxBits);
      int flag = (k < nvalid && vis.get_local(lp)) ? 1 : 0;
      int tot;
      int ex = block_escan_i32(flag, s_scan, tot);
      if (flag) {
        int j = (int)(key & (kKeyCap - 1));
        const double *q = rows5 + (int64_t)j * 5;
        int dst = n_total + base + ex, lr = n_log + base + ex;
        float4 f;
        f.x = (float)q[0];
        f.y = (float)q[1];
        f.z = (float)q[2];
        f.w = (float)q[3];
        reinterpret_cast<float4 *>(b.xyzi)[(int64_t)s * b.cap + dst] = f;
        b.label[(int64_t)s * b.cap + dst] = (uint32_t)(int64_t)q[4];
        b.pix[(int64_t)s * b.cap + dst] = global_pix(lp);
        b.tail_ref[(int64_t)s * b.log_cap + (dst - n_head)] = lr;
        double *l = b.log5 + ((int64_t)s * b.log_cap + lr) * 5;
        l[0] = q[0];
        l[1] = q[1];
        l[2] = q[2];
        l[3] = q[3];
        l[4] = q[4];
        b.log_birth[(int64_t)s * b.log_cap + lr] = step;
        if ((s_oob[j >> 5] >> (j & 31)) & 1u) *s_rebase = 1;   // bounds move: new extreme elevation
      }
      base += tot;
    }
    if (tid == 0 && ((s_ext[0] >= 0 && vis.get_local(s_ext[0])) || (s_ext[1] >= 0 && vis.get_local(s_ext[1]))))
      *s_rebase = 1;                                         // a point that holds a bound is culled
    uint16_t *stamp = b.stamp + (int64_t)s * npix;
    uint32_t *ever = b.ever + (int64_t)s * words;
    for (int ci = tid; ci < ncand; ci += kST) {
      int lp = (int)cand[ci];
      if (!vis.get_local(lp)) continue;
      int q = (int)(cand_rc[ci] >> 16) * cols + (int)(cand_rc[ci] & 0xFFFF);
      unsigned long long nv = sample_key(q, lp);
      if (nv != R3D_SENT && key_depth(nv) > R3D_EMPTY_DEPTH) {  // the pixel now holds a far return
        int f = atomicAdd(&b.n_far[s], 1);
        if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = q;
        else atomicOr(s_flags, R3D_S_FAR_OVERFLOW);
      }
      stamp[q] = (uint16_t)step;
    }
    // "this pixel was visible at some step": the visible bits of a window word are the bits of one word
    // of the scene's `ever` image (window columns are whole words), and only this workgroup writes it
    for (int e = tid; e < ww; e += kST) {
      uint32_t bits = vis.w[e];
      if (bits) ever[(win.row_of(e) * cols >> 5) + win.word_of(e)] |= bits;
    }
  }
  __syncthreads();

  STAMP(11);
  // -- 9. leave the global scratch images all-empty (only touched when LDS was too small), publish
  if (!c_lds)
    for (int e = tid; e < ww * 8; e += kST) {
      int q0 = win.row_of(e >> 3) * cols + (win.word_of(e >> 3) << 5) + ((e & 7) << 2);
      ulonglong2 sent = make_ulonglong2(R3D_SENT, R3D_SENT);
      reinterpret_cast<ulonglong2 *>(grid + q0)[0] = sent;
      reinterpret_cast<ulonglong2 *>(grid + q0)[1] = sent;
    }
  if (!s_lds)
    for (int k = tid; k < nvalid; k += kST) {
      int lp = (int)(s_keys[k] >> kIdxBits);
      if (k == 0 || (int)(s_keys[k - 1] >> kIdxBits) != lp) sgrid[global_pix(lp)] = R3D_SENT;
    }
  STAMP(12);
#ifdef R3D_STAMPS
  if (tid == 0) {
    long long *dbg = reinterpret_cast<long long *>(b.out_xyzi + (int64_t)s * b.cap * 4);
    dbg[13] = ((long long)ww << 32) | (unsigned)ncand;
    dbg[14] = ((long long)*s_nlist << 32) | (unsigned)nvalid;
  }
#endif
  if (tid == 0) {
    n_visible[s] = nvis;
    accepted[s] = accept ? 1 : 0;
    if (*s_flags) atomicOr(&b.status[s], *s_flags);
    if (accept) {
      b.n_total[s] = n_total + nvis;
      b.n_log[s] = n_log + nvis;
      if (*s_rebase) {
        b.rebase[s] += 1;                                       // single writer per scene
        if (!CHAIN) w.rebase_list[atomicAdd(w.n_rebase, 1)] = s;
      }
    }
  }
  return accept && *s_rebase != 0;        // s_rebase was last written before the barrier that ends step 8
}

__global__ void __launch_bounds__(kST)
k_insert(r3d_batch_t b, const double *__restrict__ samples5, const int64_t *__restrict__ sample_off,
         const int32_t *__restrict__ min_points, const int32_t *__restrict__ active, int step,
         int32_t *__restrict__ n_visible, int32_t *__restrict__ accepted, BatchWs w, int chunks) {
  extern __shared__ __align__(16) unsigned char smem[];
  (void)insert_scene<false>(b, samples5, sample_off, min_points, active, step, n_visible, accepted, w, chunks,
                            (int)blockIdx.x, smem);
}


// ---- compaction: drop dead points (finish, or rebase) ------------------------------------------
__global__ void __launch_bounds__(kPT)
k_alive_count(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles, int chunks) {
  __shared__ int s_a[kPT / 64];
  int cnt = *count;
  int npix = b.rows * b.cols, words = (npix + 31) / 32;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    int t0 = blockIdx.x * kTile;
    int alive = 0;
    if (t0 < n) {
      // one bit per point (a 64-bit word per wave and row) so that k_alive_write need not repeat the
      // pixel-id load and the stamp lookups.  (Skipping whole chunks whose bounding box holds no
      // visible pixel was tried: fewer bytes, but one more dependent load for the chunks that do,
      // and this kernel is latency-bound -- 0.069 ms became 0.098 ms.)
      bool flag[kPerThread];
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) {
        int i = t0 + k * kPT + threadIdx.x;
        flag[k] = i < n && point_alive(b, s, i, n_head, npix, words);
      }
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) {
        int i0 = t0 + k * kPT + (threadIdx.x & ~63);
        unsigned long long m = __ballot(flag[k]);
        if ((threadIdx.x & 63) == 0 && i0 < n) {
          w.alive_bits[(int64_t)s * chunks + (i0 >> 6)] = m;
          alive += __popcll(m);
        }
      }
    }
    if ((threadIdx.x & 63) == 0) s_a[threadIdx.x >> 6] = alive;
    __syncthreads();
    if (threadIdx.x == 0) {
      int a = 0;
      for (int v = 0; v < kPT / 64; ++v) a += s_a[v];
      w.tile_alive[(int64_t)s * tiles + blockIdx.x] = a;          // 0 for tiles beyond the cloud
    }
    __syncthreads();
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
    for (int k = 0; k < kPerThread; ++k) {                  // the alive bits k_alive_count left
      int i0 = t0 + k * kPT + (threadIdx.x & ~63);
      unsigned long long m = i0 < n ? w.alive_bits[(int64_t)s * chunks + (i0 >> 6)] : 0ull;
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
          dst[o] = src[i];
          b.out_label[(int64_t)s * b.cap + o] = b.label[(int64_t)s * b.cap + i];
        }
      }
    }
    __syncthreads();                                      // s_cnt is reused by the next scene
  }
}

// ---- rebase: one workgroup re-bases one flagged scene (rare path) --------------------------------
// Triggered when an accepted insert may have moved the elevation bounds (k_insert, step 8).  Does,
// for that scene only, what the reference does for every insert (insertion.py:373-375): forget the
// culled points, recompute the bounds, re-project every remaining point.  All phases run inside one block so
// the idle case costs one empty launch; phases are separated by a device-scope fence + barrier
// because later phases re-read what earlier ones wrote.
constexpr int kRB = 1024;

__device__ __forceinline__ void phase_sync() {
  __threadfence();
  __syncthreads();
}

__device__ __forceinline__ void rebase_scene(const r3d_batch_t &b, const BatchWs &w, int chunks, const int s,
                                             unsigned long long *s_min, unsigned long long *s_max) {
  const int tid = threadIdx.x;
  const int npix = b.rows * b.cols, words = (npix + 31) / 32;
  {
    const int n = b.n_total[s], n_head = b.n_head[s];
    int32_t *pix = b.pix + (int64_t)s * b.cap;
    // (a) entomb the dead: a point whose pixel was visible after its birth gets pixel id -1 for
    // good (the stamps that say so are about to be reset).  Nothing is moved: the input slab stays
    // as it was loaded, r3d_batch_finish drops the entombed points like any other dead point.
    for (int i = tid; i < n; i += kRB)
      if (!point_alive(b, s, i, n_head, npix, words)) pix[i] = -1;
    phase_sync();
    // (b) bounds (insertion.py:78-79) via the extreme z/r of the living
    unsigned long long lmin = ~0ull, lmax = 0ull;
    int bad = 0;
    for (int i = tid; i < n; i += kRB) {
      if (pix[i] < 0) continue;
      double x, y, z;
      load_point(b, s, i, n_head, x, y, z);
      double q = z / sqrt(x * x + y * y + z * z);
      if (!(q >= -1.0 && q <= 1.0)) bad = 1;
      else {
        unsigned long long kq = ordered_key(q);
        lmin = kq < lmin ? kq : lmin;
        lmax = kq > lmax ? kq : lmax;
      }
    }
    lmin = wave_min_u64(lmin);
    lmax = wave_max_u64(lmax);
    if ((tid & 63) == 0) {
      s_min[tid >> 6] = lmin;
      s_max[tid >> 6] = lmax;
    }
    if (bad) atomicOr(&b.status[s], R3D_S_NONFINITE);
    __syncthreads();
    if (tid == 0) {
      for (int v = 1; v < kRB / 64; ++v) {
        lmin = s_min[v] < lmin ? s_min[v] : lmin;
        lmax = s_max[v] > lmax ? s_max[v] : lmax;
      }
      double max_el = acos(ordered_key_inv(lmin)), min_el = acos(ordered_key_inv(lmax));
      b.bounds[2 * s + 0] = max_el;
      b.bounds[2 * s + 1] = min_el;
      w.q_ext[2 * s + 0] = ordered_key_inv(lmin);
      w.q_ext[2 * s + 1] = ordered_key_inv(lmax);
      b.n_far[s] = 0;
    }
    // (c) reset the visibility stamps
    for (int p = tid; p < npix; p += kRB) {
      b.stamp[(int64_t)s * npix + p] = 0;
      if (p < words) b.ever[(int64_t)s * words + p] = 0u;
    }
    phase_sync();
    // (d) re-project the living (insertion.py:74-76, :104-116): pixel ids and chunk boxes
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    int flags = 0;
    for (int i0 = 0; i0 < n; i0 += kRB) {
      int i = i0 + tid;
      BoxAcc box;
      if (i < n && pix[i] >= 0) {
        double x, y, z;
        load_point(b, s, i, n_head, x, y, z);
        pix[i] = project_point(b, s, bn, x, y, z, flags, box);
      }
      unsigned long long packed = box.wave_pack();
      int c0 = i0 + (tid & ~63);
      if ((tid & 63) == 0 && c0 < n) w.chunk_box[(int64_t)s * chunks + (c0 >> 6)] = packed;
    }
    if (tid == 0) w.n_proj[s] = n;
    if (flags) atomicOr(&b.status[s], flags);
    phase_sync();
  }
}

__global__ void __launch_bounds__(kRB)
k_rebase(r3d_batch_t b, BatchWs w, int chunks) {
  __shared__ unsigned long long s_min[kRB / 64], s_max[kRB / 64];
  const int tid = threadIdx.x;
  const int cnt = *w.n_rebase;
  for (int li = blockIdx.x; li < cnt; li += gridDim.x) rebase_scene(b, w, chunks, w.rebase_list[li], s_min, s_max);
  // the last block to leave clears the list for the next insert call
  if (tid == 0) {
    __threadfence();
    int t = atomicAdd(w.rebase_ticket, 1);
    if (t == (int)gridDim.x - 1) {
      *w.rebase_ticket = 0;
      *w.n_rebase = 0;
    }
  }
}

// ---- k_insert_chain: several insert slots of every scene in one launch ----------------------------
// The slots of ONE scene depend on each other, the scenes do not.  With one launch per slot every
// slot waits for the slowest scene of the previous one; here workgroup (slot k, scene s) only waits
// for (k-1, s), so the launch lasts as long as the slowest scene's whole chain.
//
// Hand-off (cdna_hip_programming.md, Guideline 16; correct for any placement of the workgroups on
// CUs and XCDs): the producer's waves drain their stores (s_waitcnt vmcnt(0)), the workgroup meets
// at a barrier, ONE lane executes the agent-scope release, waits again and stores the scene's
// progress word with a relaxed agent-scope atomic.  The consumer polls that one word relaxed from
// ONE lane (s_sleep between polls), then that lane executes ONE agent-scope acquire and waits, the
// workgroup meets, and only then does anybody load the scene's data (the scalar cache is dropped
// too: counters are read through it).  A device-scope fence in EVERY thread instead of one lane
// made the whole kernel 2x slower.  The progress words are zeroed by a memset node before every
// launch.
// Liveness, not correctness, leans on the dispatcher: workgroups are numbered slot-major and handed
// out in that order, so (k-1, s) is resident or done before (k, s) starts to wait.  The wait is
// bounded all the same (~2 s): on a timeout the scene is flagged (R3D_S_CHAIN_TIMEOUT), its later
// slots are skipped and the caller is told; nothing hangs and nothing is silently wrong.
constexpr int kMaxChain = 8;
struct ChainSlots {
  const double *samples5[kMaxChain];
  const int64_t *sample_off[kMaxChain];
  const int32_t *min_points[kMaxChain];
  const int32_t *active[kMaxChain];
  int32_t *n_visible[kMaxChain];
  int32_t *accepted[kMaxChain];
};

__global__ void __launch_bounds__(kST)
k_insert_chain(r3d_batch_t b, ChainSlots slots, int first_step, BatchWs w, int chunks) {
  extern __shared__ __align__(16) unsigned char smem[];     // all of the CU's LDS: no static __shared__ here
  int &s_go = reinterpret_cast<int *>(smem)[31];             // after insert_scene's counters and scan cells
  unsigned long long *s_min = reinterpret_cast<unsigned long long *>(smem + 4096), *s_max = s_min + kRB / 64;
  const int k = (int)blockIdx.x / b.B, s = (int)blockIdx.x % b.B;
  const int tid = threadIdx.x;
  if (k > 0) {
    if (tid == 0) {
      int seen = 0;                                             // progress: slots of the scene done; < 0: abandoned
      for (long long spin = 0; spin < (1ll << 21); ++spin) {
        seen = __hip_atomic_load(&w.chain_progress[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen < 0 || seen >= k) break;
        __builtin_amdgcn_s_sleep(32);
      }
      int go = seen >= k;
      if (!go && seen >= 0) atomicOr(&b.status[s], R3D_S_CHAIN_TIMEOUT);
      if (!go) __hip_atomic_store(&w.chain_progress[s], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_go = go;
      // ONE agent-scope acquire after the relaxed poll: drops this CU's stale vector-L1 lines; the wait
      // holds the barrier below until the invalidate has landed
      if (go) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (!s_go) {
      if (tid == 0) {
        slots.n_visible[k][s] = 0;
        slots.accepted[k][s] = 0;
      }
      return;
    }
    // handed-off words also travel the scalar path (uniform loads of counters): drop the scalar cache too
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  }
  bool rebase = insert_scene<true>(b, slots.samples5[k], slots.sample_off[k], slots.min_points[k], slots.active[k],
                                   first_step + k, slots.n_visible[k], slots.accepted[k], w, chunks, s, smem);
  __syncthreads();
  if (rebase) rebase_scene(b, w, chunks, s, s_min, s_max);     // rare; its phases use device-scope fences
  // publish: every storing wave drains its stores, the workgroup meets, ONE lane does the agent-scope
  // release (then waits again: the order fence -> wait -> flag matters) and stores the flag relaxed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&w.chain_progress[s], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

static int check_batch(const r3d_batch_t *b) {
  if (!b) return fail(R3D_E_ARG, "batch: null descriptor");
  if (b->B <= 0 || b->rows <= 0 || b->cols <= 0 || b->cap <= 0 || b->log_cap <= 0)
    return fail(R3D_E_ARG, "batch: non-positive shape");
  if (b->cap > (int64_t)1 << 30 || b->rows > 65535 || b->cols > 65535 || (int64_t)b->rows * b->cols > (int64_t)1 << 30)
    return fail(R3D_E_ARG, "batch: cap or range image too large for 32-bit point / pixel ids");
  if (!b->xyzi || !b->label || !b->pix || !b->n_head || !b->n_total || !b->tail_ref || !b->log5 ||
      !b->log_birth || !b->n_log || !b->grid || !b->sgrid || !b->stamp || !b->ever || !b->bounds ||
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

static size_t insert_lds_bytes(const r3d_batch_t &b) {
  (void)b;
  return (size_t)kLdsBytes;      // carved at run time: keys | bit images | sample depths | scene tile
}

// bounds -> reset -> project for the scenes of (list, count); rows = block rows of the launches.
static int launch_reproject(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                            const int32_t *count, int rows, hipStream_t st) {
  int tiles = tiles_of(b);
  hipLaunchKernelGGL(k_bounds, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w);
  int64_t npix = (int64_t)b.rows * b.cols;
  int rb = (int)((npix / 32 + kPT - 1) / kPT);
  hipLaunchKernelGGL(k_prepare, dim3(rb < 1 ? 1 : rb, rows), dim3(kPT), 0, st, b, list, count, w);
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
  hipLaunchKernelGGL(k_alive_count, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, tiles, chunks_of(b));
  hipLaunchKernelGGL(k_alive_write, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, tiles, chunks_of(b),
                     rows4, n_rows);
  R3D_LAUNCHED("compaction kernels");
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
  size_t bytes = (size_t)b->B * b->rows * b->cols * sizeof(unsigned long long);
  // the two scratch range images are all-empty between calls; every kernel leaves them so
  R3D_HIP(hipMemsetAsync(b->grid, 0xFF, bytes, st));
  R3D_HIP(hipMemsetAsync(b->sgrid, 0xFF, bytes, st));
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
  int64_t npix = (int64_t)b->rows * b->cols;
  (void)npix;
  hipLaunchKernelGGL(k_begin_init, dim3((b->B + 255) / 256), dim3(256), 0, st, *b, n_points, w);
  return launch_reproject(*b, w, w.all_list, w.all_count, b->B, st);
}

int r3d_batch_launch_one(const r3d_batch_t *b, int32_t which, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  int tiles = tiles_of(*b);
  int64_t npix = (int64_t)b->rows * b->cols;
  switch (which) {
    case R3D_K_BOUNDS:
      hipLaunchKernelGGL(k_bounds, dim3(tiles, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w);
      break;
    case R3D_K_PREPARE:
      hipLaunchKernelGGL(k_prepare, dim3((int)((npix / 32 + kPT - 1) / kPT), b->B), dim3(kPT), 0, st, *b,
                         w.all_list, w.all_count, w);
      break;
    case R3D_K_PROJECT:
      hipLaunchKernelGGL(k_project, dim3(project_blocks(*b), b->B), dim3(kPT), project_lds_bytes(*b), st, *b,
                         w.all_list, w.all_count, w, chunks_of(*b));
      break;
    case R3D_K_ALIVE_COUNT:
      hipLaunchKernelGGL(k_alive_count, dim3(tiles, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w, tiles,
                         chunks_of(*b));
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

int r3d_batch_insert(const r3d_batch_t *b, const double *samples5, const int64_t *sample_off,
                     const int32_t *min_points, const int32_t *active, int32_t step,
                     int32_t *n_visible, int32_t *accepted, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!samples5 || !sample_off || !min_points || !n_visible || !accepted || step < 1 || step > 65535)
    return fail(R3D_E_ARG, "batch_insert: null pointer or step outside [1, 65535]");
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  size_t lds = insert_lds_bytes(*b);
  static thread_local size_t lds_opted = 0;
  if (lds > lds_opted) {
    R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_opted = lds;
  }
  hipLaunchKernelGGL(k_insert, dim3(b->B), dim3(kST), lds, st, *b, samples5, sample_off, min_points,
                     active, (int)step, n_visible, accepted, w, chunks_of(*b));
  R3D_LAUNCHED("k_insert");
  // idle unless k_insert flagged a scene: then that scene is compacted and re-projected like step 0
  int rb = b->B < kRebaseRows ? b->B : kRebaseRows;
  hipLaunchKernelGGL(k_rebase, dim3(rb), dim3(kRB), 0, st, *b, w, chunks_of(*b));
  R3D_LAUNCHED("k_rebase");
  return R3D_OK;
}

int r3d_batch_insert_many(const r3d_batch_t *b, int32_t n_slots, const double *const *samples5,
                          const int64_t *const *sample_off, const int32_t *const *min_points,
                          const int32_t *const *active, int32_t first_step, int32_t *const *n_visible,
                          int32_t *const *accepted, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!samples5 || !sample_off || !min_points || !n_visible || !accepted || n_slots < 1 || first_step < 1 ||
      first_step + n_slots - 1 > 65535)
    return fail(R3D_E_ARG, "batch_insert_many: null pointer, no slot or step outside [1, 65535]");
  for (int k = 0; k < n_slots; ++k)
    if (!samples5[k] || !sample_off[k] || !min_points[k] || !n_visible[k] || !accepted[k])
      return fail(R3D_E_ARG, "batch_insert_many: null pointer in a slot");
  static const bool no_chain = getenv("R3D_NO_CHAIN") != nullptr;      // escape hatch: always one launch per slot
  if (no_chain) {
    for (int k = 0; k < n_slots; ++k) {
      rc = r3d_batch_insert(b, samples5[k], sample_off[k], min_points[k], active ? active[k] : nullptr, first_step + k,
                            n_visible[k], accepted[k], stream);
      if (rc != R3D_OK) return rc;
    }
    return R3D_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  size_t lds = insert_lds_bytes(*b);
  static thread_local size_t lds_opted = 0;
  if (lds > lds_opted) {
    R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_chain),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_opted = lds;
  }
  for (int k0 = 0; k0 < n_slots; k0 += kMaxChain) {
    int nk = n_slots - k0 < kMaxChain ? n_slots - k0 : kMaxChain;
    ChainSlots sl{};
    for (int k = 0; k < nk; ++k) {
      sl.samples5[k] = samples5[k0 + k];
      sl.sample_off[k] = sample_off[k0 + k];
      sl.min_points[k] = min_points[k0 + k];
      sl.active[k] = active ? active[k0 + k] : nullptr;
      sl.n_visible[k] = n_visible[k0 + k];
      sl.accepted[k] = accepted[k0 + k];
    }
    R3D_HIP(hipMemsetAsync(w.chain_progress, 0, (size_t)b->B * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_insert_chain, dim3(b->B * nk), dim3(kST), lds, st, *b, sl, (int)(first_step + k0), w,
                       chunks_of(*b));
    R3D_LAUNCHED("k_insert_chain");
  }
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
