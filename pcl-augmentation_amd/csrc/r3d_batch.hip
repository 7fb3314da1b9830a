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
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace r3d {

// ---- step 0 / rebase: bounds ------------------------------------------------------------------
// f64: the frame's points live in the log like inserted points do (r3d_batch_begin_f64)
__global__ void k_begin_init(r3d_batch_t b, const int32_t *n_points, BatchWs w, bool f64) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {
    *w.all_count = b.B;
  }
  if (s >= b.B) return;
  int n = n_points[s];
  int st = 0;
  if (n < 0 || n > b.cap || (f64 && n > b.log_cap)) {
    n = 0;
    st = R3D_S_CAPACITY;
  }
  b.n_head[s] = f64 ? 0 : n;
  b.n_total[s] = n;
  b.n_log[s] = f64 ? n : 0;
  b.n_far[s] = 0;
  b.rebase[s] = 0;
  b.status[s] = st;
  b.n_out[s] = 0;
  w.all_list[s] = s;
  w.shadow_valid[s] = 0;
#ifdef R3D_EXP_IMAGE
  w.img_valid[s] = 0;
#endif
  w.n_virt[s] = 0;
  w.box_area[s] = 0;
  w.qkeys[2 * s + 0] = ~0ull;   // running min of z/r
  w.qkeys[2 * s + 1] = 0ull;    // running max of z/r
}

// r3d_batch_begin_f64: rows of [x y z intensity label] float64 -> log rows 0..n-1 (exact), their float32
// rounding in xyzi / label (what the .bin writers and the compaction copy), tail_ref = identity, birth step 0.
__global__ void __launch_bounds__(kPT)
k_load_f64(r3d_batch_t b, const double *__restrict__ rows5, const int32_t *n_points) {
  const int s = blockIdx.y;
  int n = n_points[s];
  if (n < 0 || n > b.cap || n > b.log_cap) return;            // k_begin_init flags the scene
  for (int i = blockIdx.x * kPT + threadIdx.x; i < n; i += gridDim.x * kPT) {
    const double *q = rows5 + ((int64_t)s * b.cap + i) * 5;
    double v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
    double *l = b.log5 + ((int64_t)s * b.log_cap + i) * 5;
    l[0] = v0;
    l[1] = v1;
    l[2] = v2;
    l[3] = v3;
    l[4] = v4;
    b.log_birth[(int64_t)s * b.log_cap + i] = 0;
    b.tail_ref[(int64_t)s * b.log_cap + i] = i;
    reinterpret_cast<float4 *>(b.xyzi)[(int64_t)s * b.cap + i] = make_float4((float)v0, (float)v1, (float)v2, (float)v3);
    b.label[(int64_t)s * b.cap + i] = (uint32_t)(int64_t)v4;
  }
}

// r3d_batch_begin_xyz: x y z rows of 12 bytes -> the float4 slab every kernel reads (intensity 0).  Four points per thread:
// three 16-byte loads, four 16-byte stores.
__global__ void __launch_bounds__(kPT)
k_expand_xyz(r3d_batch_t b, const float *__restrict__ xyz3, const int32_t *n_points) {
  const int s = blockIdx.y;
  const int n = n_points[s];
  if (n < 0 || n > b.cap) return;                              // k_begin_init flags the scene
  const float4 *src = reinterpret_cast<const float4 *>(xyz3 + (int64_t)s * b.cap * 3);
  float4 *dst = reinterpret_cast<float4 *>(b.xyzi) + (int64_t)s * b.cap;
  const int groups = (n + 3) >> 2;                             // (cap is a multiple of 4 points: the last group stays inside the slab)
  for (int g = blockIdx.x * kPT + threadIdx.x; g < groups; g += gridDim.x * kPT) {
    const float4 a = src[3 * g], c = src[3 * g + 1], e = src[3 * g + 2];
    dst[4 * g + 0] = make_float4(a.x, a.y, a.z, 0.f);
    dst[4 * g + 1] = make_float4(a.w, c.x, c.y, 0.f);
    dst[4 * g + 2] = make_float4(c.z, c.w, e.x, 0.f);
    dst[4 * g + 3] = make_float4(e.y, e.z, e.w, 0.f);
  }
}

// elevation = acos(z/r) is monotone in q = z/r, so the bounds of insertion.py:78-79 are acos of
// the extreme q: reduce q here (sqrt + divide per point), take acos twice per scene afterwards.
//
// The float64 square root and division (~60 instructions) are only spent on points that can be an
// extreme.  k_bounds_sample first reduces the exact q of one wave-row in 32 (the first row of every
// 2048-point tile) into qkeys: the q of two real points, so the true extremes lie at or beyond them.
// k_bounds then screens every point in float32: qf = z * rsq(x*x + y*y + z*z) is within 3e-7 of z/r
// (see k_project), so a point with  lo + 2e-6 < qf < hi - 2e-6  lies strictly between two points of the
// scene and is neither extreme; it is also finite and inside [-1, 1], so it raises no flag.  Whatever
// the screen cannot exclude (the top and bottom ring of a scan; inserted float64 points; anything not
// a normal float32) takes the exact evaluation as before.
__device__ __forceinline__ void exact_q(const r3d_batch_t &b, int s, int i, int n_head, float4 pt,
                                        unsigned long long &lmin, unsigned long long &lmax, int &bad) {
  double x = (double)pt.x, y = (double)pt.y, z = (double)pt.z;
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

// One wave per 2048-point tile reads the tile's first row (64 points: one row in 32); a workgroup (4 tiles)
// leaves one pair of atomics.  (Per-wave atomics serialise: 15 000 waves on 2 x 32 addresses took 0.13 ms in
// config C5; a few workgroups per scene looping over the tiles are latency-bound, 0.024 ms.)
constexpr int kSampleHeads = 8;            // tiles whose first 64 points one wave of k_bounds_sample looks at
__global__ void __launch_bounds__(kPT)
k_bounds_sample(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w) {
  // the first 64 points of every tile; a wave takes kSampleHeads tiles and has their loads under way together (one tile per
  // wave was a round trip per workgroup: 31 000 workgroups of config C5's batch took 0.09 ms over 8 M points)
  __shared__ unsigned long long s_min[kPT / 64], s_max[kPT / 64];
  int cnt = *count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    if ((int64_t)blockIdx.x * (kPT / 64) * kSampleHeads * kTile >= n) continue;
    const float4 *src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    unsigned long long lmin = ~0ull, lmax = 0ull;
    int bad = 0;
    const int64_t first = ((int64_t)blockIdx.x * (kPT / 64) + wave) * kSampleHeads * kTile + lane;
    float4 pt[kSampleHeads];
#pragma unroll
    for (int u = 0; u < kSampleHeads; ++u) {
      const int64_t i = first + (int64_t)u * kTile;
      pt[u] = src[i < n ? i : (n > 0 ? n - 1 : 0)];
    }
#pragma unroll
    for (int u = 0; u < kSampleHeads; ++u) {
      const int64_t i = first + (int64_t)u * kTile;
      if (i < n) exact_q(b, s, (int)i, n_head, pt[u], lmin, lmax, bad);
    }
    lmin = wave_min_u64(lmin);
    lmax = wave_max_u64(lmax);
    if (lane == 0) {
      s_min[wave] = lmin;
      s_max[wave] = lmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int v = 1; v < kPT / 64; ++v) {
        lmin = s_min[v] < lmin ? s_min[v] : lmin;
        lmax = s_max[v] > lmax ? s_max[v] : lmax;
      }
      if (lmin != ~0ull) {
        atomicMin(&w.qkeys[2 * s + 0], lmin);
        atomicMax(&w.qkeys[2 * s + 1], lmax);
      }
    }
    __syncthreads();
  }
}

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
    // what the sample (or the tiles that finished before this one) found: every later value is only more extreme
    const unsigned long long k_lo = __hip_atomic_load(&w.qkeys[2 * s + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long k_hi = __hip_atomic_load(&w.qkeys[2 * s + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float lo_t = __builtin_inff(), hi_t = -__builtin_inff();    // no sample: nothing is excluded
    if (k_lo != ~0ull) {
      lo_t = (float)(ordered_key_inv(k_lo) + 2.5e-6);            // 2e-6 and the rounding of the conversion
      hi_t = (float)(ordered_key_inv(k_hi) - 2.5e-6);
    }
    // the thread's 8 float32 points are requested together (inserted float64 points, which only exist when a
    // re-based scene comes through here, are fetched from the log by exact_q)
    const float4 *src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    float4 pt[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      // x y z only, non-temporal: nothing else of the cloud is wanted here (0.078 ms against 0.086 ms alone)
      const float *f = reinterpret_cast<const float *>(src + (i < n ? i : n - 1));
      pt[k] = make_float4(__builtin_nontemporal_load(f), __builtin_nontemporal_load(f + 1), __builtin_nontemporal_load(f + 2), 0.f);
    }
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      float ssf = fmaf(pt[k].x, pt[k].x, fmaf(pt[k].y, pt[k].y, pt[k].z * pt[k].z));
      float qf = pt[k].z * __frsqrt_rn(ssf);
      const bool inside = (ssf > 1e-30f) & (ssf < 1e30f) & (qf > lo_t) & (qf < hi_t) & (i < n_head);
      if (i < n && !inside) exact_q(b, s, i, n_head, pt[k], lmin, lmax, bad);
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
      if (lmin != ~0ull) {
        atomicMin(&w.qkeys[2 * s + 0], lmin);
        atomicMax(&w.qkeys[2 * s + 1], lmax);
      }
    }
    __syncthreads();
  }
}

// After k_bounds, one block per scene: the elevation bounds (insertion.py:78-79) = acos of the extreme
// z/r, the row-edge table of the verified fast projection (k_project) and the living-point count of
// every compaction tile (all points of the frame are alive at step 0).
//
// Tables of the fast projection: a bin guessed in float32 is accepted only if the point lies
// strictly inside that bin's edges, tested on monotone images of the edges -- cos of the row edges
// against z/r, and the sign of the cross product with the unit vector of the column edges -- with a
// margin far above the rounding of either side (float32 screen, float64 for what it leaves open).
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
      // float32 screen of k_project: z/r (float32, error < 4e-7) strictly inside (c + m, c' - m) implies the
      // float64 test above with room to spare.  m = 4e-7 + d, where moving c by d moves c*|c| by more than
      // twice the float64 test's margin 4e-12: d = 1e-7 for |c| >= 1e-4 (2*|c|*d >= 2e-11), else d = 4e-6
      // (d*d/2 = 8e-12, the worst case c = d/2).  Thresholds are rounded away from the edge.
      double m = 4e-7 + (fabs(c) >= 1e-4 ? 1e-7 : 4e-6);
      float below = (float)(c - m), above = (float)(c + m);
      if ((double)below > c - m) below = nextafterf(below, -2.f);
      if ((double)above < c + m) above = nextafterf(above, 2.f);
      w.row_qf[((int64_t)s * (b.rows + 2) + k) * 2 + 0] = below;   // upper limit of z/r for the row below edge k-1
      w.row_qf[((int64_t)s * (b.rows + 2) + k) * 2 + 1] = above;   // lower limit of z/r for the row above it
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
  w.col_dirf[2 * c + 0] = (float)cos(alpha);
  w.col_dirf[2 * c + 1] = (float)sin(alpha);
}

// ---- step 0 / rebase: spherical projection -> pixel ids ------------------------------------------
// insertion.py:74-76 and :104-116 fused: r/az/el are never stored, only the pixel id of every
// point.  The min-reduce of :118-125 is NOT done here for the whole image: a scene's range image
// is only ever read in the window around an inserted object, so k_insert builds exactly that
// window from the points.  To find those points without scanning the cloud,
// every 64 consecutive points (one wave) leave their row / column bounding box; LiDAR files are
// ring-ordered, so a box is about one row by 50 columns.
// The bin (row, col) of a point is guessed in float32 and confirmed on monotone images of the bin's
// edges; a point that cannot be confirmed is queued for the reference formula (k_project_slow).
//   rows: elevation in [edge_k, edge_k+1)  <=>  cos(edge_k+1) < z/r <= cos(edge_k); both sides are
//         mapped through t -> t*|t| (strictly increasing) and multiplied by r*r = ss, which needs
//         neither the square root nor the division: z*|z| against c*|c| * ss.  Row 0 also takes
//         the truncated interval below edge_0 (int() rounds toward zero).
//   cols: the point lies counter-clockwise of column edge k and clockwise of edge k+1 (sign of
//         the cross product with the edges' unit vectors).
// Margins of the float64 confirmation are relative 4e-12 resp. 1e-12 (the L1 norm bounds r from above),
// three orders of magnitude above the rounding of the products and of the reference's own float64 evaluation.
// Cheap float32 angle guesses (about 1e-5 rad, a few per mille of a bin): whatever they get wrong the
// confirmation rejects, so their accuracy only decides how many points take the slow path, never a result.
// (guess_acosf, guess_atan2f, confirm_bin: r3d_batch.hpp -- the insert kernels bin their samples with them too)

// k_project screens in float32 first; the float64 confirmation above runs only for the points the screen
// cannot decide (a few per mille, those within ~1e-6 of a bin edge).  The screen is a sufficient condition
// for confirm_bin() on the same (rg, cg), so it never changes a result:
//   rows: qf = z * rsq(x*x + y*y + z*z) in float32 is within 2.7e-7 of z/r (three roundings of the sum,
//         halved by the root; one ulp of v_rsq_f32; one of the product; all relative, |z/r| <= 1); the
//         thresholds row_qf keep qf 4e-7 + d away from the cosine of either edge, d chosen in k_prepare so
//         that z*|z|/ss clears the float64 margin; |qf| < 0.9999 covers the pole test.
//   cols: fmaf(ax, y, -(ay * x)) with float32 table entries is within 2.4e-7 * (|x| + |y|) of the float64
//         cross product; it has to clear 1e-6 * (|x| + |y|), the float64 test needs 1e-12 * (|x|+|y|+|z|)
//         and |z| < 71 * hypot(x, y) away from the poles.
//   ss must be a normal float32 far from overflow / flush-to-zero (1e-30 < ss < 1e30).
constexpr int kVirtAreaCap = 4096;           // pixels of a chunk's box that count towards box_area (k_virt_hist)
constexpr int kVirtMeanArea = 1024;          // pixels: a scene whose mean chunk box exceeds this comes in no file order (k_virt_hist)
constexpr int kDbgVirtual = 1024;            // r3d_batch_t.reserved: every scene in virtual order (tests)
// The per-point code, counted in vector instructions (round 4: 141 per 64 points; now 69 on the usual path):
//   * the points come through a buffer descriptor per segment (base = the segment's first point, records = what is left
//     of the scene): the lane's offset is a constant, the round's a scalar, past the end reads zeros -- no address arithmetic;
//   * angle guesses and clamps in float32 (v_med3 before the conversion), octant by selects (no fmax / fmin: those
//     canonicalise their operands first);
//   * the row limits of a bin side by side in LDS ({upper, lower} at the row's index: one read, no select for row 0);
//   * the chunk's box is only reduced when the 64 points are not one ascending run of a row (lane 0's and lane 63's pixel
//     otherwise); its store address is scalar.
#ifndef R3D_PROJECT_WAVES
#define R3D_PROJECT_WAVES
#endif
#ifndef R3D_PROJECT_LOAD_AUX          // cache policy bits of the point loads / the pixel-id stores (2: non-temporal).  Same box,
#define R3D_PROJECT_LOAD_AUX 2        // one call: 0 / 0 127.0 us, loads 2: 124.8, stores 2: 136.0, both: 127.0 per 256 scenes
#endif
#ifndef R3D_PROJECT_STORE_AUX
#define R3D_PROJECT_STORE_AUX 0
#endif
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ int box_area_capped(unsigned long long packed, int cols) {
  const int r0 = (int)(packed & 0xFFFF), r1 = (int)((packed >> 16) & 0xFFFF), c0 = (int)((packed >> 32) & 0xFFFF), c1 = (int)((packed >> 48) & 0xFFFF);
  const int a = r0 > r1 ? 0 : (r1 - r0 + 1) * (c0 <= c1 ? c1 - c0 + 1 : cols - c0 + c1 + 1);
  return a > kVirtAreaCap ? kVirtAreaCap : a;
}
// The grid is what the device holds at once (project_grid(): one workgroup of 1 024 threads per CU); the scenes of (list,
// count) are cut into units of 512 consecutive points, counted from their n_total, and the units are dealt out evenly: a
// workgroup takes units [lo, hi) of that sequence, as segments (scene, first unit, last unit) it finds itself with a block
// scan of the scenes' unit counts.  It stages the column table once and a scene's row table per segment; inside a segment
// its sixteen waves take the units one at a time from a counter in LDS and need no barrier: a wave's eight rounds of 64
// points are its own (chunk boxes and alive words are per 64 points).  Measured on 256 scenes of 120 000 points:
//   * a workgroup's start (tables, scene parameters, first loads) costs 6-10 us -- as much as a tile of 2 048 points took
//     when every 4 tiles had a workgroup of their own (0.170 ms per launch; 1 / 2 / 8 tiles: 0.297 / 0.199 / 0.182);
//   * four persistent workgroups of 256 threads per CU with equal shares ended 97 / 106 / 115 / 125 us after the launch,
//     in the order the CU had received them (the older waves are served first): 0.137-0.143 ms; the waves of ONE
//     workgroup sharing the CU's range end together;
//   * what is in flight bounds the rate of a load-per-round scheme (two rounds ahead: 12.6 MB under way); a unit's eight
//     rounds are requested together, a unit ahead of their use (one descriptor per segment, the round's offset in the
//     scalar operand, past the segment's end zeros).
constexpr int kSegCap = 256;                  // units of a workgroup's range looked at per pass (segments <= units)
#ifndef R3D_PROJECT_NT
#define R3D_PROJECT_NT 1024
#endif
constexpr int kProjNT = R3D_PROJECT_NT;       // 1 024: one workgroup per CU, its sixteen waves share the work of the CU's range
constexpr int kUnit = 64 * kPerThread;        // points a wave takes at a time: eight rounds of 64 consecutive points
__global__ void __launch_bounds__(kProjNT) R3D_PROJECT_WAVES
k_project(r3d_batch_t b, const int32_t *list, const int32_t *count, int known_count, BatchWs w, int chunks) {
  extern __shared__ __align__(16) float s_tabf[];          // [(cols+1)*2] column edges, [rows*2] row limits
  __shared__ int s_scan[kProjNT / 64 + 1], s_seg[kSegCap * 4], s_nseg, s_next;
  float2 *s_col = reinterpret_cast<float2 *>(s_tabf), *s_row = s_col + (b.cols + 1);
#ifdef R3D_EXP_STAMP
  const unsigned long long stamp0 = wall_clock64();
  unsigned long long stamp1 = 0;
#endif
  for (int e = threadIdx.x; e < b.cols + 1; e += kProjNT) s_col[e] = reinterpret_cast<const float2 *>(w.col_dirf)[e];
  const int cnt = known_count >= 0 ? known_count : *count;
  const int lane = (int)threadIdx.x & 63, lane_off = lane * 16;
  const float row_top = (float)(b.rows - 1), col_top = (float)(b.cols - 1);
  // a launch of up to kProjNT scenes (the usual one): a scene per thread, one scan, nothing read twice
  const bool few = cnt <= kProjNT;
  int my_s = 0, my_n = 0, my_pre = 0, all_units;
  {
    int mine = 0;
    if (few) {
      if ((int)threadIdx.x < cnt) my_s = list[threadIdx.x], my_n = b.n_total[my_s], mine = (my_n + kUnit - 1) / kUnit;
    } else {
      for (int e = threadIdx.x; e < cnt; e += kProjNT) mine += (b.n_total[list[e]] + kUnit - 1) / kUnit;
    }
    my_pre = block_escan_i32(mine, s_scan, all_units);
  }
  const int lo = (int)((long long)blockIdx.x * all_units / gridDim.x);
  const int hi = (int)((long long)(blockIdx.x + 1) * all_units / gridDim.x);
  for (int sub = lo; sub < hi; sub += kSegCap) {
  const int sub_hi = sub + kSegCap < hi ? sub + kSegCap : hi;
  if (threadIdx.x == 0) s_nseg = 0;
  __syncthreads();
  auto offer = [&](int s_e, int n_e, int pre) {               // the part of a scene's units [pre, pre + t) inside [sub, sub_hi)
    const int t = (n_e + kUnit - 1) / kUnit;
    const int a = pre > sub ? pre : sub, z = pre + t < sub_hi ? pre + t : sub_hi;
    if (a < z) {
      const int g = atomicAdd(&s_nseg, 1);
      s_seg[4 * g + 0] = s_e, s_seg[4 * g + 1] = n_e, s_seg[4 * g + 2] = a - pre, s_seg[4 * g + 3] = z - pre;
    }
  };
  if (few) {
    offer(my_s, my_n, my_pre);
  } else {
    for (int base = 0, running = 0; base < cnt && running < sub_hi; base += kProjNT) {
      const int e = base + (int)threadIdx.x;
      const int s_e = e < cnt ? list[e] : 0, n_e = e < cnt ? b.n_total[s_e] : 0;
      int total;
      const int pre = running + block_escan_i32((n_e + kUnit - 1) / kUnit, s_scan, total);
      offer(s_e, n_e, pre);
      running += total;
    }
  }
  __syncthreads();
  const int n_seg = __builtin_amdgcn_readfirstlane(s_nseg);
  for (int g = 0; g < n_seg; ++g) {
    const int s = __builtin_amdgcn_readfirstlane(s_seg[4 * g + 0]), n = __builtin_amdgcn_readfirstlane(s_seg[4 * g + 1]);
    const int unit_lo = __builtin_amdgcn_readfirstlane(s_seg[4 * g + 2]), unit_hi = __builtin_amdgcn_readfirstlane(s_seg[4 * g + 3]);
    const int n_head = b.n_head[s];
    __syncthreads();                                       // previous scene's row table is no longer read
    {
      const float2 *rq = reinterpret_cast<const float2 *>(w.row_qf) + (int64_t)s * (b.rows + 2);
      for (int e = threadIdx.x; e < b.rows; e += kProjNT)      // (+-0.9999: the screen's pole test, folded into the limits)
        s_row[e] = make_float2(fminf(rq[e == 0 ? 0 : e + 1].x, 0.9999f), fmaxf(rq[e + 2].y, -0.9999f));
      if (threadIdx.x == 0) s_next = unit_lo;                  // the segment's units, taken by the waves one at a time
    }
    __syncthreads();
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    // diagnostic: reference formula only; a scene with float64 points (r3d_batch_begin_f64): the float32 slab gives the
    // guess, the float64 confirmation decides on the exact coordinates from the log (the float32 screen is a statement
    // about float32-exact inputs)
    const bool exact = b.reserved & 1, any64 = n_head < n;
    const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az);
    const float elo = (float)(bn.min_el + 0.00001);
    const double *row_cc = w.row_q + (int64_t)s * (b.rows + 2);
    uint32_t *queue = w.cand + (int64_t)s * w.cand_stride;  // insert scratch, free during step 0
    int flags = 0, area = 0;
    // unconfirmed points are queued for k_project_slow
    const float *scene_xyzi = b.xyzi + (int64_t)s * b.cap * 4;   // (the slab holds the float32 rounding of float64 points)
    const int first = unit_lo * kUnit, left = n - first, span = (unit_hi - unit_lo) * kUnit;
    const __amdgpu_buffer_rsrc_t seg_xyzi = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(scene_xyzi + (int64_t)first * 4), 0, (left < span ? left : span) * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t seg_pix = __builtin_amdgcn_make_buffer_rsrc(
        b.pix + (int64_t)s * b.cap + first, 0, (left < span ? left : span) * 4, 0x00020000);
    // (past the segment's end the loads return zeros: the rounds requested ahead at its last unit)
#ifdef R3D_EXP_L2
    auto fetch = [&](int round) { return __builtin_amdgcn_raw_buffer_load_b96(seg_xyzi, lane_off, (round & 7) * 1024, 0); };
#else
    auto fetch = [&](int round) { return __builtin_amdgcn_raw_buffer_load_b96(seg_xyzi, lane_off, round * 1024, R3D_PROJECT_LOAD_AUX); };
#endif
    // One round of a wave: 64 consecutive points, one per lane.  round_any: the general form (lanes past the scene's end, float64 points, the
    // diagnostic mode).
    auto round_any = [&](const int t0, const int round0, const int k, const u32x3 pt) {
      const int i = t0 + k * 64 + lane;
      const float px = __uint_as_float(pt.x), py = __uint_as_float(pt.y), pz = __uint_as_float(pt.z);
      int row = 0, col = 0;
      bool placed = false;
      const bool live = i < n;
      if (live) {
        const float ssf = fmaf(px, px, fmaf(py, py, pz * pz));
        float qf = pz * __frsqrt_rn(ssf);
        qf = __builtin_amdgcn_fmed3f(qf, -1.f, 1.f);
        row = (int)floorf((guess_acosf(qf) - elo) * inv_del);
        col = (int)((guess_atan2f(py, px) + 3.14159274f) * inv_daz);
        row = max(0, min(row, bn.rows - 1));
        col = max(0, min(col, bn.cols - 1));
        const float2 ea = s_col[col], eb = s_col[col + 1], rq = s_row[row];
        const float mcf = 1e-6f * (fabsf(px) + fabsf(py));
        int ok = (int)(ssf > 1e-30f) & (int)(ssf <= 249000.f) & (int)(qf < rq.x) & (int)(qf > rq.y) &
                 (int)(fmaf(ea.x, py, -(ea.y * px)) > mcf) & (int)(fmaf(eb.x, py, -(eb.y * px)) < -mcf) & (int)(!any64);
        bool far = false;
        if (!ok) {                                           // undecided in float32, or r near / above 500
          double x = (double)px, y = (double)py, z = (double)pz;
          if (i >= n_head) load_point(b, s, i, n_head, x, y, z);
          double ss = x * x + y * y + z * z;
          ok = confirm_bin(row_cc, w.col_dir, row, col, x, y, z, ss);
          far = ss > R3D_EMPTY_DEPTH * R3D_EMPTY_DEPTH;      // r > 500 (or rounds to it): far list
        }
        if (ok & (int)(!exact)) {
          int p = (int)pack_pix(row, col);
          placed = true;
          if (far) {
            int f = atomicAdd(&b.n_far[s], 1);
            if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = p;
            else flags |= R3D_S_FAR_OVERFLOW;
          }
          __builtin_amdgcn_raw_buffer_store_b32((unsigned int)p, seg_pix, lane_off >> 2, (round0 + k) * 256, R3D_PROJECT_STORE_AUX);
        } else {
          queue[atomicAdd(&w.n_slow[s], 1)] = (uint32_t)i;
        }
      }
      // the chunk's box.  A scan in ring order gives 64 points of one row whose columns rise with the lane:
      // then the box is lane 0's and lane 63's pixel (checked, not assumed); anything else takes the reduction.
      unsigned long long packed;
      {
        const int before = __builtin_amdgcn_update_dpp(col, col, 0x138, 0xF, 0xF, false);   // wave_shr:1; lane 0 keeps its own
        const int row0 = __builtin_amdgcn_readfirstlane(row);
        const bool in_order = placed & (row == row0) & (col >= before);
        if (__ballot(in_order) == ~0ull) {
          packed = pack_box(row0, row0, __builtin_amdgcn_readfirstlane(col), __builtin_amdgcn_readlane(col, 63));
        } else {
          BoxAcc box;
          if (live) {
            if (placed) {
              box.add(row, col);
            } else {                                           // unknown pixel (queued for k_project_slow): the chunk's
              box.add(0, 0);                                   // box covers the whole image until k_fix_boxes
              box.add(b.rows - 1, b.cols - 1);
            }
          }
          packed = __ballot(live & !placed) ? box.wave_pack() : box.wave_pack_arc(placed ? col : -1, b.cols);
        }
      }
      const int i0 = t0 + k * 64;                              // the wave's first point (scalar)
      if (i0 < n) {
        if ((k & 3) == 0) area += 4 * box_area_capped(packed, b.cols);
        const unsigned long long living = __ballot(live);     // every point of the frame is alive at step 0
        if ((threadIdx.x & 63) == 0) {
          w.chunk_box[(int64_t)s * chunks + (i0 >> 6)] = packed;
          w.alive[(int64_t)s * chunks + (i0 >> 6)] = living;
        }
      }
    };
    // round_whole: every lane has a point and the scene is plain float32.  No "is there a point" mask; the float32 screen
    // and the test for an ascending run of one row are ONE chain of v_cmpx (each narrows EXEC, none needs a scalar AND;
    // EXEC is put back inside the statement); everything about the lanes the screen leaves undecided sits behind one
    // uniform branch.
    auto round_whole = [&](const int t0, const int round0, const int k, const u32x3 pt) {
      const float px = __uint_as_float(pt.x), py = __uint_as_float(pt.y), pz = __uint_as_float(pt.z);
      // bin guess (about 1e-5 rad off at worst, a few per mille of a bin)
      const float ssf = fmaf(px, px, fmaf(py, py, pz * pz));
      float qf = pz * __frsqrt_rn(ssf);
      qf = __builtin_amdgcn_fmed3f(qf, -1.f, 1.f);
      const float rowf = floorf((guess_acosf(qf) - elo) * inv_del);
      const float ax = fabsf(px), ay = fabsf(py);
      const bool steep = ay > ax;                              // octant by selects (fmax / fmin canonicalise their operands first)
      const float mx = steep ? ay : ax, mn = steep ? ax : ay;
      const float t = mn * __builtin_amdgcn_rcpf(mx), t2 = t * t;   // atan on [0, 1], odd polynomial (v_rcp_f32: 1 ulp)
      float pa = fmaf(-0.01172120f, t2, 0.05265332f);
      pa = fmaf(pa, t2, -0.11643287f);
      pa = fmaf(pa, t2, 0.19354346f);
      pa = fmaf(pa, t2, -0.33262347f);
      pa = fmaf(pa, t2, 0.99997726f);
      float az = pa * t;
      az = steep ? 1.57079637f - az : az;
      az = px < 0.f ? 3.14159274f - az : az;
      az = py < 0.f ? -az : az;
      const int row = (int)__builtin_amdgcn_fmed3f(rowf, 0.f, row_top);
      const int col = (int)__builtin_amdgcn_fmed3f((az + 3.14159274f) * inv_daz, 0.f, col_top);
      const float2 ea = s_col[col], eb = s_col[col + 1], rq = s_row[row];
      const float mcf = 1e-6f * (ax + ay);
      const float c1 = fmaf(ea.x, py, -(ea.y * px)), c2 = fmaf(eb.x, py, -(eb.y * px));
      const int before = __builtin_amdgcn_update_dpp(col, col, 0x138, 0xF, 0xF, false);   // wave_shr:1; lane 0 keeps its own
      const int row0 = __builtin_amdgcn_readfirstlane(row);
      // (ss up to 249 000: r below 499; the row limits in LDS stay within +-0.9999: the pole test)
      unsigned long long ok_mask, run_mask, saved;
      asm volatile("s_mov_b64 %[sv], exec\n\t"
                   "v_cmpx_lt_f32_e32 vcc, 0x0da24260, %[ss]\n\t"          // 1e-30 < ss
                   "v_cmpx_ge_f32_e32 vcc, 0x48732a00, %[ss]\n\t"          // 249 000 >= ss
                   "v_cmpx_gt_f32_e32 vcc, %[hi], %[q]\n\t"
                   "v_cmpx_lt_f32_e32 vcc, %[lo], %[q]\n\t"
                   "v_cmpx_gt_f32_e32 vcc, %[c1], %[m]\n\t"
                   "v_cmpx_lt_f32_e64 vcc, %[c2], -%[m]\n\t"
                   "s_mov_b64 %[ok], exec\n\t"
                   "v_cmpx_eq_u32_e32 vcc, %[row0], %[row]\n\t"
                   "v_cmpx_le_i32_e32 vcc, %[bef], %[col]\n\t"
                   "s_mov_b64 %[run], exec\n\t"
                   "s_mov_b64 exec, %[sv]\n\t"
                   "s_nop 2"                                             // (a DPP operation may follow: 5 states after the last v_cmpx)
                   : [ok] "=&s"(ok_mask), [run] "=&s"(run_mask), [sv] "=&s"(saved)
                   : [ss] "v"(ssf), [q] "v"(qf), [hi] "v"(rq.x), [lo] "v"(rq.y), [c1] "v"(c1), [c2] "v"(c2), [m] "v"(mcf),
                     [row0] "s"(row0), [row] "v"(row), [bef] "v"(before), [col] "v"(col)
                   : "vcc");
      const int p = (int)pack_pix(row, col);
      unsigned long long packed, placed_mask = ~0ull;
      if (ok_mask == ~0ull) {
#ifndef R3D_EXP_NOSTORE
        __builtin_amdgcn_raw_buffer_store_b32((unsigned int)p, seg_pix, lane_off >> 2, (round0 + k) * 256, R3D_PROJECT_STORE_AUX);
#endif
      } else {                                                 // some lane undecided in float32, or r near / above 500
        bool ok = __builtin_amdgcn_inverse_ballot_w64(ok_mask), far = false;
        if (!ok) {
          const double x = (double)px, y = (double)py, z = (double)pz, ss = x * x + y * y + z * z;
          ok = confirm_bin(row_cc, w.col_dir, row, col, x, y, z, ss);
          far = ss > R3D_EMPTY_DEPTH * R3D_EMPTY_DEPTH;        // r > 500 (or rounds to it): far list
        }
        if (ok) {
          if (far) {
            int f = atomicAdd(&b.n_far[s], 1);
            if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = p;
            else flags |= R3D_S_FAR_OVERFLOW;
          }
          __builtin_amdgcn_raw_buffer_store_b32((unsigned int)p, seg_pix, lane_off >> 2, (round0 + k) * 256, R3D_PROJECT_STORE_AUX);
        } else {
          queue[atomicAdd(&w.n_slow[s], 1)] = (uint32_t)(t0 + k * 64 + lane);
        }
        placed_mask = __ballot(ok);
      }
      if (run_mask == ~0ull) {                                 // (all decided by the screen, one row, columns ascending)
        packed = pack_box(row0, row0, __builtin_amdgcn_readfirstlane(col), __builtin_amdgcn_readlane(col, 63));
      } else {
        const bool placed = __builtin_amdgcn_inverse_ballot_w64(placed_mask);
        BoxAcc box;
        if (placed) {
          box.add(row, col);
        } else {                                               // unknown pixel (queued for k_project_slow): the chunk's
          box.add(0, 0);                                       // box covers the whole image until k_fix_boxes
          box.add(b.rows - 1, b.cols - 1);
        }
        packed = placed_mask != ~0ull ? box.wave_pack() : box.wave_pack_arc(col, b.cols);
      }
      // the boxes' areas (k_virt_hist: is this a scene whose points come in no file order?): every fourth round's
      if ((k & 3) == 0) area += 4 * box_area_capped(packed, b.cols);
      if ((threadIdx.x & 63) == 0) {
        const int c = (t0 >> 6) + k;
        w.chunk_box[(int64_t)s * chunks + c] = packed;
        w.alive[(int64_t)s * chunks + c] = ~0ull;              // every point of the frame is alive at step 0
      }
    };
    const bool plain = !any64 && !exact;
#ifdef R3D_EXP_STAMP
    if (!stamp1) stamp1 = wall_clock64();
#endif
    // A wave takes the segment's units one at a time from the counter in LDS (the sixteen waves of the CU finish together:
    // four workgroups of 256 threads with their own ranges ended 97 / 106 / 115 / 125 us after the launch, in the order
    // the CU had received them -- the older waves are served first); the next unit is taken, and its eight rounds are
    // requested, before the current one is worked on.
    auto claim = [&]() {
      int u = 0;
      if (lane == 0) u = atomicAdd(&s_next, 1);
      return __builtin_amdgcn_readfirstlane(u);
    };
    int unit = claim();
    if (unit < unit_hi) {
      u32x3 cur[kPerThread], nxt[kPerThread];
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) cur[k] = fetch((unit - unit_lo) * kPerThread + k);
      while (unit < unit_hi) {
        const int unit_next = claim();
        // (past the segment's end the loads return zeros)
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) nxt[k] = fetch((unit_next - unit_lo) * kPerThread + k);
        const int t0 = unit * kUnit, round0 = (unit - unit_lo) * kPerThread;
        if (plain && t0 + kUnit <= n) {
#pragma unroll
          for (int k = 0; k < kPerThread; ++k) round_whole(t0, round0, k, cur[k]);
        } else {
#pragma unroll
          for (int k = 0; k < kPerThread; ++k) round_any(t0, round0, k, cur[k]);
        }
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) cur[k] = nxt[k];
        unit = unit_next;
      }
    }
    flags = wave_or_i32(flags);
    if ((threadIdx.x & 63) == 0 && flags) atomicOr(&b.status[s], flags);
    // how large are this scene's chunk boxes?  (k_virt_hist: a scene whose points come in no file order)
    if ((threadIdx.x & 63) == 0 && area) atomicAdd(&w.box_area[s], area);
  }
  }
#ifdef R3D_EXP_STAMP
  if (threadIdx.x == 0) {
    unsigned long long *o = reinterpret_cast<unsigned long long *>(b.out_xyzi) + (size_t)blockIdx.x * 4;
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned int hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    o[0] = stamp0, o[1] = stamp1, o[2] = wall_clock64(), o[3] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
}

// The reference formula (insertion.py:74-76, :104-116) for the points k_project could not confirm:
// none to a handful per scan (points within 1e-12 of a bin edge, or a float32 guess one bin off).
__global__ void __launch_bounds__(kPT)
k_project_slow(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w) {
  int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n_slow = w.n_slow[s], n_head = b.n_head[s];
    if ((b.reserved & R3D_B_FILE_ORDER) && !(b.reserved & kDbgVirtual) && blockIdx.x == 0 && threadIdx.x == 0) {
      const int n = b.n_total[s];                            // (k_project has left the boxes' areas: was the promise kept?)
      if (n >= 4096 && (long long)w.box_area[s] > (long long)kVirtMeanArea * ((n + 63) >> 6)) atomicAdd(&w.dbg[kCntPromiseBroken], 1);
    }
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    const uint32_t *queue = w.cand + (int64_t)s * w.cand_stride;
    int flags = 0, n_risk = 0;
    for (int e = blockIdx.x * kPT + threadIdx.x; e < n_slow; e += gridDim.x * kPT) {
      int i = (int)queue[e];
      double x, y, z;
      load_point(b, s, i, n_head, x, y, z);
      BoxAcc unused;
      b.pix[(int64_t)s * b.cap + i] = project_point(b, s, bn, x, y, z, flags, unused, &n_risk);
    }
    if (flags) atomicOr(&b.status[s], flags);
    if (n_risk) atomicAdd(&w.dbg[kCntEdgeScene], n_risk);       // (rare: a handful per billion points)
  }
}

// ... and the boxes of their chunks.  k_project gave a chunk with an unconfirmed point the whole image as box (the point's
// pixel was not known yet); on a grid several times finer than the reference's a per cent or two of the points go that
// way, i.e. a few per cent of the CHUNKS -- some 600 of a 1M-point scan on 448 x 2880, every one of them in every
// insert's chunk list.  One wave per queued point (duplicates of a chunk write the same box): the chunk's 64 final
// pixels, their box with the columns as the shorter arc.
__global__ void __launch_bounds__(kPT)
k_fix_boxes(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int chunks) {
  int cnt = *count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    const int n_slow = w.n_slow[s], n = b.n_total[s];
    if (b.status[s] & (R3D_S_ROW_RANGE | R3D_S_COL_RANGE | R3D_S_NONFINITE)) continue;   // (a point without a pixel: the boxes stay whole)
    const uint32_t *queue = w.cand + (int64_t)s * w.cand_stride;
    for (int e = blockIdx.x * (kPT / 64) + wave; e < n_slow; e += gridDim.x * (kPT / 64)) {
      const int c = (int)(queue[e] >> 6), i = (c << 6) + lane;
      BoxAcc box;
      const uint32_t p = i < n ? (uint32_t)b.pix[(int64_t)s * b.cap + i] : 0u;
      if (i < n) box.add(pix_row(p), pix_col(p));
      const unsigned long long packed = box.wave_pack_arc(i < n ? pix_col(p) : -1, b.cols);
      if (lane == 0) w.chunk_box[(int64_t)s * chunks + c] = packed;
    }
  }
}

// The super-boxes of a freshly projected scene: one wave per 64 chunks, a chunk per lane.
__global__ void __launch_bounds__(kPT)
k_super_rows(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int chunks) {
  int cnt = *count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_sup = (chunks + 63) >> 6;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n_chunks = (b.n_total[s] + 63) >> 6;
    for (int sp = blockIdx.x * (kPT / 64) + wave; sp < n_sup; sp += gridDim.x * (kPT / 64)) {
      const int c = (sp << 6) + lane;
      int r0 = 0x7FFFFFFF, r1 = -1;
      if (c < n_chunks) {
        const unsigned long long bx = w.chunk_box[(int64_t)s * chunks + c];
        const int a = (int)(bx & 0xFFFF), z = (int)((bx >> 16) & 0xFFFF);
        if (a <= z) r0 = a, r1 = z;                            // (an empty box: rows 0xFFFF .. 0)
      }
      r0 = wave_min_i32(r0);
      r1 = wave_max_i32(r1);
      if (lane == 0) {
        w.super_rows[((int64_t)s * n_sup + sp) * 2 + 0] = r0;
        w.super_rows[((int64_t)s * n_sup + sp) * 2 + 1] = r1;
      }
    }
  }
}

// ---- virtual order (r3d_batch.hpp): a cloud whose points come in no LiDAR file order ---------------------------------
// Is the mean box of a scene's chunks large?  A scan in ring order has boxes of one row by ~50 columns, a firing-sequence
// order (all lasers of one azimuth, then the next) one column by every row; a shuffled cloud has the whole image in every
// box, every chunk then sits in every insert's list and an insert costs a pass over the cloud (config C2 shuffled, round 5
// before this: 1.51 ms per launch of 5 slots against 0.32).  k_project leaves the sum of the boxes' areas (each capped at
// kVirtAreaCap pixels: a chunk with a point on the slow path has the whole image as box until k_fix_boxes) in box_area[s].
// If so: a counting sort of the point NUMBERS by (group of rows, band of columns) -- at most kVirtBins bins of a few hundred
// points each, finer ones than a chunk would buy nothing -- in three streaming kernels: k_virt_hist (a histogram per block
// of kVirtBlock points), k_virt_scan (counts -> every block's first place per bin), k_virt_scatter ({pixel,
// point} to its place, inv), k_virt_finish (pixel ids and perm in the new order, the chunk boxes over it).  The order inside
// a bin is whatever the atomics give: nothing that is computed afterwards depends on it (minima per pixel, kills per pixel,
// the output in slab order).  (First form, one 1024-thread workgroup per scene with the bins in LDS: 1.42 ms per 256 scenes
// of 120 000 points -- more than the inserts of config C2 gain.)
constexpr int kVirtBins = 512;
#ifndef R3D_VIRT_PER
#define R3D_VIRT_PER 32
#endif
constexpr int kVirtPer = R3D_VIRT_PER;       // points per thread of the histogram / scatter kernels, taken 16 at a time
constexpr int kVirtGo = 16;
constexpr int kVirtBlock = kPT * kVirtPer;   // points per block of the histogram / scatter kernels
struct VirtShape {
  int mode, gh, cshift, nbands, nbins, nblk;
};
inline int virt_blocks(const r3d_batch_t &b) { return (int)((b.cap + kVirtBlock - 1) / kVirtBlock); }
__device__ __forceinline__ int virt_bin(const VirtShape &v, uint32_t p) { return (pix_row(p) / v.gh) * v.nbands + (pix_col(p) >> v.cshift); }

// does the scene get a virtual order?  (the same answer in k_virt_hist and k_virt_scan)
__device__ __forceinline__ bool virt_wanted(const r3d_batch_t &b, const BatchWs &w, const VirtShape &v, int s, int n) {
  if (b.status[s] & (R3D_S_ROW_RANGE | R3D_S_COL_RANGE | R3D_S_NONFINITE)) return false;   // (a point without a pixel)
  if (v.mode == 2 || (b.reserved & kDbgVirtual)) return n > 0;
  return n >= 4096 && (long long)w.box_area[s] > (long long)kVirtMeanArea * ((n + 63) >> 6);
}

__global__ void __launch_bounds__(kPT)
k_virt_hist(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, VirtShape v) {
  __shared__ uint32_t s_hist[kVirtBins];
  const int tid = threadIdx.x;
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n = b.n_total[s];
    const int blk = (int)blockIdx.x;
    if (blk * kVirtBlock >= n || !virt_wanted(b, w, v, s, n)) continue;
    __syncthreads();
    for (int e = tid; e < v.nbins; e += kPT) s_hist[e] = 0u;
    __syncthreads();
    const int32_t *pix = b.pix + (int64_t)s * b.cap;
    for (int u0 = 0; u0 < kVirtPer; u0 += kVirtGo) {
      uint32_t p[kVirtGo];
#pragma unroll
      for (int u = 0; u < kVirtGo; ++u) {
        const int i = blk * kVirtBlock + (u0 + u) * kPT + tid;
        p[u] = i < n ? (uint32_t)pix[i] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int u = 0; u < kVirtGo; ++u)
        if (p[u] != 0xFFFFFFFFu) atomicAdd(&s_hist[virt_bin(v, p[u])], 1u);
    }
    __syncthreads();
    uint32_t *off = w.sort_off + ((int64_t)s * v.nblk + blk) * kVirtBins;
    for (int e = tid; e < v.nbins; e += kPT) off[e] = s_hist[e];
  }
}

// counts -> every block's first place per bin (bin-major: all of bin 0, then bin 1, ...); one block per scene.  (A kernel of
// its own: the scene's last histogram block doing this needs an agent-scope release per block -- an L2 write-back on this
// part, 1 us each and one after the other per XCD: 0.9 ms for the 7 680 blocks of config C2.)
constexpr int kVirtScanNT = 512;
__global__ void __launch_bounds__(kVirtScanNT)
k_virt_scan(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, VirtShape v) {
  static_assert(kVirtBins <= kVirtScanNT, "a bin per thread");
  __shared__ int s_scan[kVirtScanNT / 64 + 1];
  const int tid = threadIdx.x;
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n = b.n_total[s], nblk = (n + kVirtBlock - 1) / kVirtBlock;
    if (!virt_wanted(b, w, v, s, n)) continue;
    uint32_t *all = w.sort_off + (int64_t)s * v.nblk * kVirtBins;
    int total = 0;
    if (tid < v.nbins) {
#pragma unroll 8
      for (int bl = 0; bl < nblk; ++bl) total += (int)all[(int64_t)bl * kVirtBins + tid];
    }
    int tot;
    int run = block_escan_i32(total, s_scan, tot);
    if (tid < v.nbins) {
#pragma unroll 8
      for (int bl = 0; bl < nblk; ++bl) {
        const int c = (int)all[(int64_t)bl * kVirtBins + tid];
        all[(int64_t)bl * kVirtBins + tid] = (uint32_t)run;
        run += c;
      }
    }
    if (tid == 0) {
      w.n_virt[s] = n;
      atomicAdd(&w.dbg[kCntVirtual], 1);
    }
  }
}

__global__ void __launch_bounds__(kPT)
k_virt_scatter(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, VirtShape v) {
  // A block's points go to their bins THROUGH LDS: placed there in bin order (local counts, their scan, a cursor per bin),
  // then streamed out -- the entries of a bin are neighbours in LDS and in the scene's sorted array, so the stores of a wave
  // cover whole runs (scattered straight from the lanes, 8 bytes each, the kernel took 0.22 ms per 256 scenes: three times
  // what its bytes cost).  inv[] is written where the point number is the lane's (coalesced).
  extern __shared__ __align__(16) unsigned char s_virt[];
  uint2 *s_stage = reinterpret_cast<uint2 *>(s_virt);                      // [kVirtBlock]
  uint32_t *s_goff = reinterpret_cast<uint32_t *>(s_stage + kVirtBlock);   // [kVirtBins] the block's first place per bin
  uint32_t *s_lstart = s_goff + kVirtBins, *s_cur = s_lstart + kVirtBins;  // the same inside the block | cursor
  __shared__ int s_scan[kPT / 64 + 1];
  static_assert(kVirtBins == 2 * kPT, "two bins per thread in the scan of the block's counts");
  const int tid = threadIdx.x;
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n = w.n_virt[s];
    const int blk = (int)blockIdx.x;
    if (blk * kVirtBlock >= n) continue;                     // (n_virt = 0: the scene keeps its order)
    __syncthreads();
    const uint32_t *off = w.sort_off + ((int64_t)s * v.nblk + blk) * kVirtBins;
    // the block's count per bin = the distance to the next place in the scene's order (this bin in the next block; behind
    // the scene's last block the next bin in the first; behind everything n): k_virt_scan left exactly these places
    {
      const int nblk_s = (n + kVirtBlock - 1) / kVirtBlock;
      const uint32_t *first = w.sort_off + (int64_t)s * v.nblk * kVirtBins;
      for (int e = tid; e < kVirtBins; e += kPT) {
        uint32_t at = 0u, next = 0u;
        if (e < v.nbins) {
          at = off[e];
          next = blk + 1 < nblk_s ? off[kVirtBins + e] : (e + 1 < v.nbins ? first[e + 1] : (uint32_t)n);
        }
        s_goff[e] = at;
        s_cur[e] = next - at;
      }
    }
    const int32_t *pix = b.pix + (int64_t)s * b.cap;
    uint2 *tmp = reinterpret_cast<uint2 *>(b.out_xyzi) + (int64_t)s * b.cap * 2;   // (the output slab: scratch until r3d_batch_finish)
    uint32_t *inv = w.inv + (int64_t)s * b.cap;
    __syncthreads();
    {
      const int c0 = (int)s_cur[2 * tid], c1 = (int)s_cur[2 * tid + 1];
      int total;
      const int ex = block_escan_i32(c0 + c1, s_scan, total);
      s_lstart[2 * tid] = s_cur[2 * tid] = (uint32_t)ex;
      s_lstart[2 * tid + 1] = s_cur[2 * tid + 1] = (uint32_t)(ex + c0);
    }
    __syncthreads();
    for (int u0 = 0; u0 < kVirtPer; u0 += kVirtGo) {         // into LDS in bin order
      uint32_t p[kVirtGo];
#pragma unroll
      for (int u = 0; u < kVirtGo; ++u) {
        const int i = blk * kVirtBlock + (u0 + u) * kPT + tid;
        p[u] = i < n ? (uint32_t)pix[i] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int u = 0; u < kVirtGo; ++u) {
        if (p[u] == 0xFFFFFFFFu) continue;
        const int i = blk * kVirtBlock + (u0 + u) * kPT + tid;
        const int bin = virt_bin(v, p[u]);
        const uint32_t l = atomicAdd(&s_cur[bin], 1u);
        s_stage[l] = make_uint2(p[u], (uint32_t)i);
        inv[i] = s_goff[bin] + (l - s_lstart[bin]);
      }
    }
    __syncthreads();
    const int here = n - blk * kVirtBlock < kVirtBlock ? n - blk * kVirtBlock : kVirtBlock;
    for (int l = tid; l < here; l += kPT) {                  // ... and out
      const uint2 e = s_stage[l];
      const int bin = virt_bin(v, e.x);
      tmp[s_goff[bin] + ((uint32_t)l - s_lstart[bin])] = e;
    }
  }
}

__global__ void __launch_bounds__(kPT)
k_virt_finish(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int chunks) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n = w.n_virt[s];
    const int t0 = (int)blockIdx.x * kTile;
    if (t0 >= n) continue;
    int32_t *pix = b.pix + (int64_t)s * b.cap;
    const uint2 *tmp = reinterpret_cast<const uint2 *>(b.out_xyzi) + (int64_t)s * b.cap * 2;
    uint32_t *perm = w.perm + (int64_t)s * b.cap;
    uint2 e[kPerThread];
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int j = t0 + u * kPT + tid;
      e[u] = j < n ? tmp[j] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int j = t0 + u * kPT + tid;
      BoxAcc box;
      if (j < n) {
        pix[j] = (int32_t)e[u].x;
        perm[j] = e[u].y;
        box.add(pix_row(e[u].x), pix_col(e[u].x));
      }
      const unsigned long long packed = box.wave_pack_arc(j < n ? pix_col(e[u].x) : -1, b.cols);
      if (lane == 0 && (t0 + u * kPT + (tid & ~63)) < n) w.chunk_box[(int64_t)s * chunks + (j >> 6)] = packed;
    }
  }
}

// A scene in virtual order: its alive bits back in slab order (what the compaction, the delta and the float64 rows are made
// from), and the living points per 2048-point tile of the slabs.  One block per tile; the other scenes are left alone.
// Two launches: every original point alive (the appended ones keep their own bits: they are numbered alike in both orders),
// then the dead points of the virtual order clear their bit in slab order -- a few per cent of the points, found by walking
// the alive words once.  (Looked up the other way round, every point through inv[], the kernel took 0.10 ms per 256 scenes.)
template <bool ROWS4>
__global__ void __launch_bounds__(kPT)
k_unvirtual_all(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles, int chunks) {
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n_virt = w.n_virt[s];
    if (!n_virt) continue;
    const int n = b.n_total[s];
    const bool shadow = ROWS4 && w.shadow_valid[s] != 0;     // (what k_alive_write<ROWS4> shows)
    const unsigned long long *alive = (shadow ? w.alive_shadow : w.alive) + (int64_t)s * chunks;
    for (int c0 = blockIdx.x * kPT; (c0 << 6) < n; c0 += gridDim.x * kPT) {
      const int c = c0 + (int)threadIdx.x, lo = c << 6;
      if (lo < n) {
        const unsigned long long in_scene = n - lo >= 64 ? ~0ull : (1ull << (n - lo)) - 1ull;
        const unsigned long long orig = lo >= n_virt ? 0ull : (n_virt - lo >= 64 ? ~0ull : (1ull << (n_virt - lo)) - 1ull);
        const unsigned long long word = (orig | (orig != ~0ull ? alive[c] & ~orig : 0ull)) & in_scene;
        w.alive_o[(int64_t)s * chunks + c] = word;
      }
    }
  }
}

template <bool ROWS4>
__global__ void __launch_bounds__(kPT)
k_unvirtual_dead(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles, int chunks) {
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n_virt = w.n_virt[s];
    if (!n_virt) continue;
    const bool shadow = ROWS4 && w.shadow_valid[s] != 0;
    const unsigned long long *alive = (shadow ? w.alive_shadow : w.alive) + (int64_t)s * chunks;
    const uint32_t *perm = w.perm + (int64_t)s * b.cap;
    // a word of the virtual order per lane; the dead points of the wave's 64 words are listed in LDS (their place among the
    // wave's 4 096 points, by the lanes' prefix sums), then taken 64 at a time, one per lane: one trip to perm[] per 64 dead
    // points (walked lane by lane, or word by word, the trips followed one another: 71 / 56 us per 256 scenes)
    __shared__ uint16_t s_dead[kPT / 64][4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = blockIdx.x * kPT + (int)(threadIdx.x & ~63); (c0 << 6) < n_virt; c0 += gridDim.x * kPT) {
      const int c = c0 + lane, lo = c << 6;
      unsigned long long dead = 0ull;
      if (lo < n_virt) {
        const unsigned long long valid = n_virt - lo >= 64 ? ~0ull : (1ull << (n_virt - lo)) - 1ull;
        dead = ~alive[c] & valid;
      }
      const int mine = __popcll(dead);
      const int incl = wave_iscan_i32(mine), total = wave_last_i32(incl);
      if (!total) continue;
      int at = incl - mine;
      while (dead) {
        const int bit = __ffsll((long long)dead) - 1;
        dead &= dead - 1;
        s_dead[wave][at++] = (uint16_t)((lane << 6) | bit);
      }
      __builtin_amdgcn_wave_barrier();
      for (int e = lane; e < total; e += 64) {
        const uint32_t i = perm[(c0 << 6) + (int)s_dead[wave][e]];
        atomicAnd(&w.alive_o[(int64_t)s * chunks + (i >> 6)], ~(1ull << (i & 63)));
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ... and the living points per 2 048-point tile of the slabs, from the finished words (a count kept by the dead points
// themselves was one more atomic each).
__global__ void __launch_bounds__(kPT)
k_unvirtual_tiles(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles, int chunks) {
  static_assert(kTile == 32 * 64, "a tile is 32 alive words");
  const int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    if (!w.n_virt[s]) continue;
    const int n = b.n_total[s];
    for (int c = blockIdx.x * kPT + (int)threadIdx.x; ((c & ~63) << 6) < n; c += gridDim.x * kPT) {
      int living = (c << 6) < n ? __popcll(w.alive_o[(int64_t)s * chunks + c]) : 0;
      living += __shfl_xor(living, 1, 64);                     // the 32 words of a tile: half a wave
      living += __shfl_xor(living, 2, 64);
      living += __shfl_xor(living, 4, 64);
      living += __shfl_xor(living, 8, 64);
      living += __shfl_xor(living, 16, 64);
      if ((threadIdx.x & 31) == 0 && (c >> 5) < tiles && ((int64_t)(c >> 5) * kTile) < n) w.tile_o[(int64_t)s * tiles + (c >> 5)] = living;
    }
  }
}

static void launch_unvirtual(const r3d_batch_t &b, const BatchWs &w, const int32_t *list, const int32_t *count, int rows, bool rows4,
                             hipStream_t st) {
  const int tiles = tiles_of(b), chunks = chunks_of(b);
  const dim3 g_all((chunks + kPT - 1) / kPT, rows), g_dead((chunks + kPT - 1) / kPT, rows);
  if (rows4) {
    hipLaunchKernelGGL((k_unvirtual_all<true>), g_all, dim3(kPT), 0, st, b, list, count, w, tiles, chunks);
    hipLaunchKernelGGL((k_unvirtual_dead<true>), g_dead, dim3(kPT), 0, st, b, list, count, w, tiles, chunks);
  } else {
    hipLaunchKernelGGL((k_unvirtual_all<false>), g_all, dim3(kPT), 0, st, b, list, count, w, tiles, chunks);
    hipLaunchKernelGGL((k_unvirtual_dead<false>), g_dead, dim3(kPT), 0, st, b, list, count, w, tiles, chunks);
  }
  hipLaunchKernelGGL(k_unvirtual_tiles, g_all, dim3(kPT), 0, st, b, list, count, w, tiles, chunks);
}

// Clouds of many tiles (config C5: 489 per scan): the living points in front of every tile, once per scene, instead of every
// block of the compaction adding up the counts of the tiles before it (up to 8 dependent loads per lane at its start).
constexpr int kPrefixMinTiles = 128;
template <bool ROWS4>
__global__ void __launch_bounds__(1024)
k_tile_prefix(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles) {
  __shared__ int s_scan[1024 / 64 + 1];
  __shared__ int s_carry;
  const int cnt = *count, tid = threadIdx.x;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const bool shadow = ROWS4 && w.shadow_valid[s] != 0, virt = w.n_virt[s] != 0;
    const int32_t *src = (virt ? w.tile_o : (shadow ? w.tile_shadow : w.tile_alive)) + (int64_t)s * tiles;
    const int used = (b.n_total[s] + kTile - 1) / kTile;
    __syncthreads();
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < used; base += 1024) {
      const int t = base + tid;
      const int c = t < used ? src[t] : 0;
      int tot;
      const int ex = block_escan_i32(c, s_scan, tot);
      const int carry = s_carry;
      if (t < used) w.tile_pre[(int64_t)s * tiles + t] = carry + ex;
      __syncthreads();
      if (tid == 0) s_carry = carry + tot;
      __syncthreads();
    }
  }
}

// SceneBatch.pixel_ids(): the pixel id of every point in slab order as the reference numbers it (row * cols + col)
__global__ void k_export_pix(r3d_batch_t b, BatchWs w, int32_t *out) {
  const int s = blockIdx.y;
  const int n = b.n_total[s], n_virt = w.n_virt[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = i < n_virt ? (int)w.inv[(int64_t)s * b.cap + i] : i;
    const uint32_t p = (uint32_t)b.pix[(int64_t)s * b.cap + j];
    out[(int64_t)s * b.cap + i] = pix_row(p) * b.cols + pix_col(p);
  }
}

// Survivors in original order (insertion.py:472-473 applied once for all steps), float4 + label
// straight into the output arrays.  A wave owns 8 consecutive chunks (512 points) of its block's tile and
// needs nobody else: its output offset is the sum of the living counts of the scene's preceding tiles
// (tile_alive, kept by the inserts) and of the preceding chunks of its own tile (popcounts of their alive
// words); the 8 alive words travel through SGPRs (v_readlane), ranks are v_mbcnt of the word, the 8 points
// and labels of a lane are requested together.  No LDS, no barrier, 32-bit offsets from scalar bases.
// (Measured and dropped: the block-wide form with ranks through an LDS table -- 93 VGPRs, 0.28 ms.)
constexpr int kWaveChunks = kTile / 64 / (kPT / 64);    // chunks of a tile per wave

template <bool ROWS4, bool NONTEMP>
__global__ void __launch_bounds__(kPT)
k_alive_write(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles, int chunks,
              double *rows4, int32_t *n_rows, bool slab_order_made) {
  static_assert(kWaveChunks == 8 && kTile / 64 <= 64, "a wave reads its tile's alive words with one load");
  int cnt = *count;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    const int s = list[li];
    const int n = b.n_total[s];
    const int t0 = blockIdx.x * kTile;
    if (t0 >= n) continue;
    // (r3d_batch_export_rows shows the copy a rejected candidate has left, while there is one: BatchWs::shadow_valid)
    const bool shadow = ROWS4 && w.shadow_valid[s] != 0;
    const bool virt = w.n_virt[s] != 0;                       // (virtual order: the bits k_unvirtual has put back into slab order)
    if (virt && !slab_order_made) {
      // the bits in slab order were never made (alive_o / tile_o hold whatever the last batch left): nothing of this scene
      // is written -- offsets taken from them would point anywhere -- and the scene is flagged
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicOr(&b.status[s], R3D_S_ORDER_PROMISE);
        (ROWS4 ? n_rows : b.n_out)[s] = 0;
      }
      continue;
    }
    const int32_t *tile_alive = (virt ? w.tile_o : (shadow ? w.tile_shadow : w.tile_alive)) + (int64_t)s * tiles;
    const unsigned long long *alive = (virt ? w.alive_o : (shadow ? w.alive_shadow : w.alive)) + (int64_t)s * chunks;
    int pre = 0;
    if (tiles >= kPrefixMinTiles) pre = lane == 0 ? w.tile_pre[(int64_t)s * tiles + blockIdx.x] : 0;   // (k_tile_prefix)
    else
      for (int t = lane; t < (int)blockIdx.x; t += 64) pre += tile_alive[t];
    const int first = wave * kWaveChunks;                    // my first chunk within the tile
    unsigned long long m = 0ull;                             // lane c: alive word of chunk c of the tile
    if (lane < first + kWaveChunks && t0 + lane * 64 < n) m = alive[(t0 >> 6) + lane];
    if (lane < first) pre += __popcll(m);
    int run = wave_sum_i32(pre);
    if (threadIdx.x == 0 && t0 + kTile >= n)                 // the scene's last tile publishes the total
      (ROWS4 ? n_rows : b.n_out)[s] = run + tile_alive[blockIdx.x];
    const int n_head = b.n_head[s];
    const float4 *__restrict__ src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    float4 *__restrict__ dst = reinterpret_cast<float4 *>(b.out_xyzi) + (int64_t)s * b.cap;
    const uint32_t *__restrict__ lsrc = b.label + (int64_t)s * b.cap;
    uint32_t *__restrict__ ldst = b.out_label + (int64_t)s * b.cap;
    const unsigned int ibase = (unsigned int)(t0 + first * 64 + lane);
    unsigned int mlo[kWaveChunks], mhi[kWaveChunks];
#pragma unroll
    for (int k = 0; k < kWaveChunks; ++k) {
      mlo[k] = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)m, first + k);
      mhi[k] = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(m >> 32), first + k);
    }
    if (ROWS4) {                                             // r3d_batch_export_rows: x y z label as float64
#pragma unroll 1
      for (int k = 0; k < kWaveChunks; ++k) {
        unsigned long long mk = ((unsigned long long)mhi[k] << 32) | mlo[k];
        if ((mk >> lane) & 1ull) {
          int i = (int)(ibase + k * 64), o = run + (int)__builtin_amdgcn_mbcnt_hi(mhi[k], __builtin_amdgcn_mbcnt_lo(mlo[k], 0u));
          double x, y, z;
          load_point(b, s, i, n_head, x, y, z);
          double2 *row = reinterpret_cast<double2 *>(rows4 + ((int64_t)s * b.cap + o) * 4);
          row[0] = make_double2(x, y);
          row[1] = make_double2(z, (double)(lsrc[i] & 0xFFFFu));
        }
        run += __popcll(mk);
      }
    } else {
      // (named scalars, not arrays: the compiler keeps arrays of float4 filled under a branch in scratch)
#ifdef R3D_CHECK
#define R3D_CHECK_O(o) if (o >= (unsigned int)b.cap) { atomicAdd(&w.dbg[15], 1); o = 0u; }   /* (code 3 of the insert kernels shares the cell) */
#else
#define R3D_CHECK_O(o)
#endif
#define R3D_LOAD(K)                                                                                     \
  unsigned int i##K = ibase + K * 64;                                                                      \
  i##K = i##K < (unsigned int)n ? i##K : (unsigned int)n - 1u; /* dead points are few: every lane loads */ \
  float4 p##K;                                                                                          \
  uint32_t l##K;                                                                                        \
  if (NONTEMP) {                                                                                        \
    const float *f = reinterpret_cast<const float *>(src + i##K);                                       \
    p##K = make_float4(__builtin_nontemporal_load(f), __builtin_nontemporal_load(f + 1),                \
                       __builtin_nontemporal_load(f + 2), __builtin_nontemporal_load(f + 3));           \
    l##K = __builtin_nontemporal_load(lsrc + i##K);                                                     \
  } else {                                                                                              \
    p##K = src[i##K];                                                                                   \
    l##K = lsrc[i##K];                                                                                  \
  }
#define R3D_STORE(K)                                                                                    \
  if ((((unsigned long long)mhi[K] << 32 | mlo[K]) >> lane) & 1ull) {                                   \
    unsigned int o = (unsigned int)run + __builtin_amdgcn_mbcnt_hi(mhi[K], __builtin_amdgcn_mbcnt_lo(mlo[K], 0u)); \
    R3D_CHECK_O(o)                                                                                        \
    if (NONTEMP) {                                                                                      \
      float *f = reinterpret_cast<float *>(dst + o);                                                    \
      __builtin_nontemporal_store(p##K.x, f);                                                           \
      __builtin_nontemporal_store(p##K.y, f + 1);                                                       \
      __builtin_nontemporal_store(p##K.z, f + 2);                                                       \
      __builtin_nontemporal_store(p##K.w, f + 3);                                                       \
      __builtin_nontemporal_store(l##K, ldst + o);                                                      \
    } else {                                                                                            \
      dst[o] = p##K;                                                                                    \
      ldst[o] = l##K;                                                                                   \
    }                                                                                                   \
  }                                                                                                     \
  run += __popc(mlo[K]) + __popc(mhi[K]);
      R3D_LOAD(0) R3D_LOAD(1) R3D_LOAD(2) R3D_LOAD(3) R3D_LOAD(4) R3D_LOAD(5) R3D_LOAD(6) R3D_LOAD(7)
      R3D_STORE(0) R3D_STORE(1) R3D_STORE(2) R3D_STORE(3) R3D_STORE(4) R3D_STORE(5) R3D_STORE(6) R3D_STORE(7)
#undef R3D_LOAD
#undef R3D_STORE
    }
  }
}

// r3d_batch_adopt_rejected: the copy a rejected candidate has left becomes the scene (the reference's driver goes on
// with it when no further candidate restores the backup: SS insertion.py:453, :468-471, then :373 / save_data).  One
// workgroup per scene: alive bits and tile counts from the shadow, then the bounds and every living point's pixel
// afresh, as the next fill_spherical / geometrical_front_view would (a rebase that was not needed changes nothing).
constexpr int kAdoptNT = 1024;
__global__ void __launch_bounds__(kAdoptNT)
k_adopt_shadow(r3d_batch_t b, BatchWs w, const int32_t *active, int chunks) {
  __shared__ unsigned long long s_min[kAdoptNT / 64], s_max[kAdoptNT / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (!w.shadow_valid[s] || (active && !active[s])) return;
  const int tiles = (int)((b.cap + kTile - 1) / kTile), n_chunks = (b.n_total[s] + 63) >> 6;
  for (int c = tid; c < n_chunks; c += kAdoptNT) w.alive[(int64_t)s * chunks + c] = w.alive_shadow[(int64_t)s * chunks + c];
  for (int t = tid; t < tiles; t += kAdoptNT) w.tile_alive[(int64_t)s * tiles + t] = w.tile_shadow[(int64_t)s * tiles + t];
  phase_sync();
  rebase_scene<kAdoptNT>(b, w, chunks, s, s_min, s_max);
  if (tid == 0) {
    w.shadow_valid[s] = 0;
    b.rebase[s] += 1;
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

// r3d_batch_export_delta: what a host that still holds the frames needs to write the merged files -- the alive word
// of every 64-point chunk and the inserted points (float32 rounding + label) in insertion order, at a fixed stride.
__global__ void k_export_delta(r3d_batch_t b, BatchWs w, int chunks, unsigned long long *alive_out, float *tail_xyzi,
                               uint32_t *tail_label, int64_t tail_stride, int32_t *counts, bool slab_order_made) {
  const int s = blockIdx.y;
  const int n_head = b.n_head[s], n_total = b.n_total[s];
  const int n_tail = n_total - n_head;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts[s] = n_head;
    counts[b.B + s] = n_total;
    if (n_tail > tail_stride) atomicOr(&b.status[s], R3D_S_CAPACITY);
    if (w.n_virt[s] != 0 && !slab_order_made) atomicOr(&b.status[s], R3D_S_ORDER_PROMISE);
  }
  const int n_words = (n_total + 63) >> 6;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += gridDim.x * blockDim.x) {
    unsigned long long a = c < n_words ? (w.n_virt[s] ? w.alive_o : w.alive)[(int64_t)s * chunks + c] : 0ull;
    const int left = n_total - (c << 6);
    if (left < 64) a = left > 0 ? a & ((1ull << left) - 1ull) : 0ull;
    alive_out[(int64_t)s * chunks + c] = a;
  }
  const float4 *src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap + n_head;
  float4 *dst = reinterpret_cast<float4 *>(tail_xyzi) + (int64_t)s * tail_stride;
  const int n_copy = n_tail < tail_stride ? n_tail : (int)tail_stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_copy; i += gridDim.x * blockDim.x) {
    dst[i] = src[i];
    tail_label[(int64_t)s * tail_stride + i] = b.label[(int64_t)s * b.cap + n_head + i];
  }
}

// r3d_batch_export_alive: the alive words in slab order, masked to the scene's count
__global__ void k_export_alive(r3d_batch_t b, BatchWs w, int chunks, unsigned long long *alive_out, bool slab_order_made) {
  const int s = blockIdx.y;
  const int n_total = b.n_total[s];
  const int n_words = (n_total + 63) >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0 && w.n_virt[s] != 0 && !slab_order_made) atomicOr(&b.status[s], R3D_S_ORDER_PROMISE);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += gridDim.x * blockDim.x) {
    // (the cloud r3d_batch_export_rows shows: the copy a rejected candidate has left while there is one)
    const unsigned long long *src = w.n_virt[s] ? w.alive_o : (w.shadow_valid[s] ? w.alive_shadow : w.alive);
    unsigned long long a = c < n_words ? src[(int64_t)s * chunks + c] : 0ull;
    const int left = n_total - (c << 6);
    if (left < 64) a = left > 0 ? a & ((1ull << left) - 1ull) : 0ull;
    alive_out[(int64_t)s * chunks + c] = a;
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
  if (((size_t)(b->cols + 1) + b->rows + 2) * 2 * sizeof(float) > 60 * 1024)          // (+ 4 KB of its own: 64 KB per workgroup)
    return fail(R3D_E_ARG, "batch: range image too large for the projection kernel's LDS edge tables");
  if ((int64_t)b->rows * b->cols >= (1 << 24))
    return fail(R3D_E_ARG, "batch: range image of 2^24 pixels or more");
  if (b->workspace_bytes < carve_batch(*b, nullptr).total)
    return fail(R3D_E_WORKSPACE, "batch: workspace smaller than r3d_batch_workspace_bytes()");
  return R3D_OK;
}

static size_t project_lds_bytes(const r3d_batch_t &b) {
  return ((size_t)(b.cols + 1) + b.rows + 2) * 2 * sizeof(float);
}

// k_project's grid: the workgroups the device holds at once (they deal the tiles out among themselves), but no more than
// there can be tiles.  (The table below is a cache of a device attribute, per device and LDS size -- not state of a batch.)
static int project_grid(const r3d_batch_t &b) {
  static std::mutex mu;
  static std::map<std::pair<int, size_t>, int> known;      // (device, LDS bytes) -> resident workgroups
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t lds = project_lds_bytes(b);
  int resident;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find({dev, lds});
    if (it == known.end()) {
      int cus = 0, per_cu = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_project, kProjNT, lds) != hipSuccess || per_cu <= 0) per_cu = 1;
      it = known.emplace(std::make_pair(dev, lds), cus * per_cu).first;
    }
    resident = it->second;
  }
  // (a small batch: no more workgroups than have a unit for every wave -- but one per scene when the scenes are that short:
  // a workgroup's segments follow one another)
  const long long units = (long long)b.B * ((b.cap + kUnit - 1) / kUnit);
  long long most = (units + kProjNT / 64 - 1) / (kProjNT / 64);
  if (most < b.B) most = b.B < units ? b.B : units;
  return (int)(most < resident ? (most > 0 ? most : 1) : resident);
}

// Can a scene of this batch be in virtual order?  0: no -- bit 2048 of `reserved` (R3D_B_FILE_ORDER): the caller says that the
// clouds come in a LiDAR file order, nothing is looked at, nothing sorted (a cloud that does not keep the promise costs time,
// not results); 1: the scenes whose chunk boxes say that their points come in no file order; 2: all (bit 1024, tests).
// finish / export_delta / export_rows skip the launches that put alive bits back into slab order when the mode is 0; a scene
// that IS in virtual order then (the caller set the bit between a begin that looked and its finish) is flagged
// R3D_S_ORDER_PROMISE by the kernel that would have read the missing bits, instead of writing a wrong cloud silently.
static int virtual_order_mode(const r3d_batch_t &b) {
  if (b.reserved & kDbgVirtual) return 2;
  return (b.reserved & R3D_B_FILE_ORDER) ? 0 : 1;
}

// bounds -> tables -> project for the scenes of (list, count); rows = block rows of the launches.
static int launch_reproject(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                            const int32_t *count, int rows, hipStream_t st, int slow_blocks = 4) {
  int tiles = tiles_of(b);
  hipLaunchKernelGGL(k_bounds_sample, dim3((tiles + kSampleHeads * (kPT / 64) - 1) / (kSampleHeads * (kPT / 64)), rows), dim3(kPT), 0, st, b,
                     list, count, w);
  hipLaunchKernelGGL(k_bounds, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w);
  hipLaunchKernelGGL(k_prepare, dim3(1, rows), dim3(kPT), 0, st, b, list, count, w, tiles);
  // (known_count: the list is all_list of a batch that has just begun -- its count is B, no need to wait for the word)
  hipLaunchKernelGGL(k_project, dim3(project_grid(b)), dim3(kProjNT), project_lds_bytes(b), st, b, list, count,
                     list == w.all_list ? b.B : -1, w, chunks_of(b));
  // (the queue of a scene grows with its size: a 1M-point scan on 448 x 2880 leaves ~600 chunks to k_fix_boxes, one wave each)
  const int fix_blocks = slow_blocks > tiles / 32 ? slow_blocks : tiles / 32;
  hipLaunchKernelGGL(k_project_slow, dim3(slow_blocks, rows), dim3(kPT), 0, st, b, list, count, w);
  hipLaunchKernelGGL(k_fix_boxes, dim3(fix_blocks, rows), dim3(kPT), 0, st, b, list, count, w, chunks_of(b));
  {
    const int mode = virtual_order_mode(b);
    if (mode) {
      VirtShape v;
      v.mode = mode;
      v.cshift = 6;                                            // bands of 64 columns, groups of rows: at most kVirtBins bins
      while (((b.cols + (1 << v.cshift) - 1) >> v.cshift) > kVirtBins) ++v.cshift;
      v.nbands = (b.cols + (1 << v.cshift) - 1) >> v.cshift;
      int groups = kVirtBins / v.nbands;
      groups = groups > b.rows ? b.rows : (groups < 1 ? 1 : groups);
      v.gh = (b.rows + groups - 1) / groups;
      v.nbins = ((b.rows + v.gh - 1) / v.gh) * v.nbands;
      v.nblk = virt_blocks(b);
      hipLaunchKernelGGL(k_virt_hist, dim3(v.nblk, rows), dim3(kPT), 0, st, b, list, count, w, v);
      hipLaunchKernelGGL(k_virt_scan, dim3(1, rows), dim3(kVirtScanNT), 0, st, b, list, count, w, v);
      {
        const size_t lds = (size_t)kVirtBlock * sizeof(uint2) + 3 * kVirtBins * sizeof(uint32_t);
        // (per device, every call: the attribute belongs to the current device's copy of the kernel)
        R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_virt_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        hipLaunchKernelGGL(k_virt_scatter, dim3(v.nblk, rows), dim3(kPT), lds, st, b, list, count, w, v);
      }
      hipLaunchKernelGGL(k_virt_finish, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, chunks_of(b));
    }
  }
  if (supers_on(b, chunks_of(b)))
    hipLaunchKernelGGL(k_super_rows, dim3((supers_of(b) + kPT / 64 - 1) / (kPT / 64), rows), dim3(kPT), 0, st, b, list, count, w,
                       chunks_of(b));
  R3D_LAUNCHED("reproject kernels");
  return R3D_OK;
}

static int launch_compact(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                          const int32_t *count, int rows, hipStream_t st, double *rows4 = nullptr,
                          int32_t *n_rows = nullptr) {
  int tiles = tiles_of(b);
  dim3 grid(tiles, rows), blk(kPT);
  // non-temporal loads and stores: nothing of the cloud is read again before r3d_batch_begin overwrites the state
  // (scenes in virtual order: their alive bits back in slab order first; a block of the others returns at once)
  if (rows4) {
    if (virtual_order_mode(b)) launch_unvirtual(b, w, list, count, rows, true, st);
    if (tiles >= kPrefixMinTiles) hipLaunchKernelGGL((k_tile_prefix<true>), dim3(1, rows), dim3(1024), 0, st, b, list, count, w, tiles);
    hipLaunchKernelGGL((k_alive_write<true, false>), grid, blk, 0, st, b, list, count, w, tiles, chunks_of(b), rows4, n_rows,
                       virtual_order_mode(b) != 0);
  } else {
    if (virtual_order_mode(b)) launch_unvirtual(b, w, list, count, rows, false, st);
    if (tiles >= kPrefixMinTiles) hipLaunchKernelGGL((k_tile_prefix<false>), dim3(1, rows), dim3(1024), 0, st, b, list, count, w, tiles);
    hipLaunchKernelGGL((k_alive_write<false, true>), grid, blk, 0, st, b, list, count, w, tiles, chunks_of(b), rows4, n_rows,
                       virtual_order_mode(b) != 0);
  }
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
  R3D_HIP(hipMemsetAsync(w.dbg, 0, 64 * sizeof(int32_t), st));
  R3D_LAUNCHED("k_col_table");
  return R3D_OK;
}

int r3d_batch_begin(const r3d_batch_t *b, const int32_t *n_points, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!n_points) return fail(R3D_E_ARG, "batch_begin: null n_points");
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  hipLaunchKernelGGL(k_begin_init, dim3((b->B + 255) / 256), dim3(256), 0, st, *b, n_points, w, false);
  return launch_reproject(*b, w, w.all_list, w.all_count, b->B, st);
}

int r3d_batch_begin_xyz(const r3d_batch_t *b, const float *xyz3, const int32_t *n_points, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!n_points || !xyz3) return fail(R3D_E_ARG, "batch_begin_xyz: null xyz3 or n_points");
  if (b->cap % 4 != 0) return fail(R3D_E_ARG, "batch_begin_xyz: cap must be a multiple of 4 points");
  if (((uintptr_t)xyz3 & 15) != 0) return fail(R3D_E_ARG, "batch_begin_xyz: xyz3 must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  int gx = (int)((b->cap / 4 + kPT * 4 - 1) / (kPT * 4));
  hipLaunchKernelGGL(k_expand_xyz, dim3(gx < 1 ? 1 : gx, b->B), dim3(kPT), 0, st, *b, xyz3, n_points);
  hipLaunchKernelGGL(k_begin_init, dim3((b->B + 255) / 256), dim3(256), 0, st, *b, n_points, w, false);
  return launch_reproject(*b, w, w.all_list, w.all_count, b->B, st);
}

int r3d_batch_begin_f64(const r3d_batch_t *b, const double *rows5, const int32_t *n_points, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!n_points || !rows5) return fail(R3D_E_ARG, "batch_begin_f64: null rows5 or n_points");
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  hipLaunchKernelGGL(k_load_f64, dim3(32, b->B), dim3(kPT), 0, st, *b, rows5, n_points);
  hipLaunchKernelGGL(k_begin_init, dim3((b->B + 255) / 256), dim3(256), 0, st, *b, n_points, w, true);
  // every point takes the reference formula: more workgroups per scene for k_project_slow
  return launch_reproject(*b, w, w.all_list, w.all_count, b->B, st, 32);
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
      hipLaunchKernelGGL(k_project, dim3(project_grid(*b)), dim3(kProjNT), project_lds_bytes(*b), st, *b,
                         w.all_list, w.all_count, b->B, w, chunks_of(*b));
      break;
    case R3D_K_ALIVE_WRITE:
      return launch_compact(*b, w, w.all_list, w.all_count, b->B, st);
#ifdef R3D_EXP_IMAGE
    case 6:
      return launch_image_clear(*b, w, st);
    case 7:
      return launch_image_build(*b, w, st);
    case 8:
      return launch_image_bands(*b, w, st);
#endif
    default:
      return fail(R3D_E_ARG, "batch_launch_one: unknown kernel id");
  }
  R3D_LAUNCHED("batch_launch_one");
  return R3D_OK;
}

int r3d_batch_adopt_rejected(const r3d_batch_t *b, const int32_t *active, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  BatchWs w = carve_batch(*b, b->workspace);
  hipLaunchKernelGGL(k_adopt_shadow, dim3(b->B), dim3(kAdoptNT), 0, (hipStream_t)stream, *b, w, active, chunks_of(*b));
  R3D_LAUNCHED("k_adopt_shadow");
  return R3D_OK;
}

int r3d_batch_export_rows(const r3d_batch_t *b, double *rows4, int32_t *n_rows, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!rows4 || !n_rows) return fail(R3D_E_ARG, "batch_export_rows: null output");
  BatchWs w = carve_batch(*b, b->workspace);
  return launch_compact(*b, w, w.all_list, w.all_count, b->B, (hipStream_t)stream, rows4, n_rows);
}

int r3d_batch_point_order(const r3d_batch_t *b, int32_t *virtual_order, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!virtual_order) return fail(R3D_E_ARG, "batch_point_order: null output");
  BatchWs w = carve_batch(*b, b->workspace);
  R3D_HIP(hipMemcpyAsync(virtual_order, w.n_virt, (size_t)b->B * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return R3D_OK;
}

int r3d_batch_export_pix(const r3d_batch_t *b, int32_t *pix_ids, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!pix_ids) return fail(R3D_E_ARG, "batch_export_pix: null output");
  BatchWs w = carve_batch(*b, b->workspace);
  hipLaunchKernelGGL(k_export_pix, dim3(32, b->B), dim3(256), 0, (hipStream_t)stream, *b, w, pix_ids);
  R3D_LAUNCHED("k_export_pix");
  return R3D_OK;
}

int r3d_batch_export_alive(const r3d_batch_t *b, uint64_t *alive, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!alive) return fail(R3D_E_ARG, "batch_export_alive: null output");
  BatchWs w = carve_batch(*b, b->workspace);
  if (virtual_order_mode(*b)) launch_unvirtual(*b, w, w.all_list, w.all_count, b->B, true, (hipStream_t)stream);
  hipLaunchKernelGGL(k_export_alive, dim3(8, b->B), dim3(256), 0, (hipStream_t)stream, *b, w, chunks_of(*b),
                     reinterpret_cast<unsigned long long *>(alive), virtual_order_mode(*b) != 0);
  R3D_LAUNCHED("k_export_alive");
  return R3D_OK;
}

int r3d_batch_export_delta(const r3d_batch_t *b, uint64_t *alive, float *tail_xyzi, uint32_t *tail_label, int64_t tail_stride,
                           int32_t *counts, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  if (!alive || !tail_xyzi || !tail_label || !counts || tail_stride <= 0) return fail(R3D_E_ARG, "batch_export_delta: null output or stride");
  BatchWs w = carve_batch(*b, b->workspace);
  if (virtual_order_mode(*b)) launch_unvirtual(*b, w, w.all_list, w.all_count, b->B, false, (hipStream_t)stream);
  hipLaunchKernelGGL(k_export_delta, dim3(8, b->B), dim3(256), 0, (hipStream_t)stream, *b, w, chunks_of(*b),
                     reinterpret_cast<unsigned long long *>(alive), tail_xyzi, tail_label, tail_stride, counts,
                     virtual_order_mode(*b) != 0);
  R3D_LAUNCHED("k_export_delta");
  return R3D_OK;
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
