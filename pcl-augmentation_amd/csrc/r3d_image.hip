// Level 2, the IMAGE ROUTE of the insert step (round 6): a persistent raw range image per scene.
//
// The chain route (r3d_insert.hip) rebuilds, for every (slot, scene) pair, the window of the scene's range image around the
// inserted object from the points: chunk list -> pixel ids -> coordinates -> minima, then one kill mask per listed chunk.
// That is work in proportion to the POINTS near the window, and it leans on the point order (chunk boxes).  Here the image
// of insertion.py:118-125 exists once per scene, in HBM:
//   img[pixel]    minimum squared depth over the LIVING points of the pixel (float64 bits; R3D_SENT: nobody)
//   occ           its occupancy bits (the label image, :119-120)
//   kstep[pixel]  step of the latest accepted insert for which the pixel was visible
// and three facts of DESIGN.md par.3 make it exact: pixel ids only change with the elevation bounds; culling is per pixel
// (:470-473) -- a point is dead iff its pixel turned visible at a step after the point's birth --; after an accepted
// insert a visible pixel holds exactly the sample's points that fell into it (or nobody: a closing-filled hole of the
// sample, :467 on the smoothed images).  An insert is then work in proportion to the SAMPLE: the occupancy words of its
// window, depth reads at its candidate pixels, one image write per visible pixel.  The alive words the rest of Level 2
// lives on (compaction, delta, float64 rows, the chain route, rebase) are brought up to date from kstep by one streaming
// pass at the end of a launch (k_apply_kills).  No step depends on the order of the points.
//
// This file: the image's construction (k_image_clear, k_image_build), k_apply_kills, the insert kernel of the route
// (k_insert_image: one workgroup per scene walks the scene's slots in order -- nobody speculates, parks or hands over) and
// the host side that chooses between the routes (launch_slots_image, called from r3d_insert.hip).
#include "r3d_insert_core.hpp"

namespace r3d {

// ---- construction ------------------------------------------------------------------------------------------------------
// Scenes whose image is not valid: everything empty.  One block row per scene, 16-byte stores.
__global__ void __launch_bounds__(kPT)
k_image_clear(r3d_batch_t b, BatchWs w) {
  const int s = blockIdx.y;
  if (w.img_valid[s]) return;
  const size_t npix = (size_t)b.rows * b.cols;
  // img: npix * 8 bytes of ones; kstep: npix * 2 bytes of zeros; occ: npix / 8 bytes of zeros (npix is a multiple of 32)
  uint4 *img = reinterpret_cast<uint4 *>(w.img + (size_t)s * npix);
  uint4 *ks = reinterpret_cast<uint4 *>(w.kstep + (size_t)s * npix);
  uint32_t *oc = w.occ + (size_t)s * (npix / 32);
  const size_t n_img = npix / 2, n_ks = npix / 8, n_oc = npix / 32;
  const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u), zeros = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = (size_t)blockIdx.x * kPT + threadIdx.x; i < n_img; i += (size_t)gridDim.x * kPT) img[i] = ones;
  for (size_t i = (size_t)blockIdx.x * kPT + threadIdx.x; i < n_ks; i += (size_t)gridDim.x * kPT) ks[i] = zeros;
  for (size_t i = (size_t)blockIdx.x * kPT + threadIdx.x; i < n_oc; i += (size_t)gridDim.x * kPT) oc[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) w.n_hold[s] = 0;
}

// One atomic OR per run of lanes that share an occupancy word (a scan in ring order: 64 consecutive points fall into two
// or three words).  `word` < 0: the lane has nothing to set.  Whole wave.
__device__ __forceinline__ void or_occ_by_runs(uint32_t *occ, int word, uint32_t bits) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int w2 = __shfl_up(word, d, 64);
    const uint32_t b2 = (uint32_t)__shfl_up((int)bits, d, 64);
    if (lane >= d && w2 == word) bits |= b2;
  }
  const int next = __shfl_down(word, 1, 64);
  if (word >= 0 && (lane == 63 || next != word)) atomicOr(&occ[word], bits);
}

// The LIVING points of the scenes whose image is not valid, min-reduced on the squared depth (insertion.py:118-125); the
// pixels of the points that hold an elevation bound (first / last row only, unless the elevation span is tiny) are noted:
// when such a point dies the bounds may move (insertion.py:373 recomputes them for every insert).
// A block takes one 2048-point tile; the points are numbered as pix / alive number them (virtual order: perm).
__global__ void __launch_bounds__(kPT)
k_image_build(r3d_batch_t b, BatchWs w, int chunks) {
  const int s = blockIdx.y;
  if (w.img_valid[s]) return;
  const int n = b.n_total[s], n_head = b.n_head[s], n_virt = w.n_virt[s];
  const int t0 = blockIdx.x * kTile;
  if (t0 >= n) return;
  const size_t npix = (size_t)b.rows * b.cols;
  unsigned long long *img = w.img + (size_t)s * npix;
  uint32_t *occ = w.occ + (size_t)s * (npix / 32);
  const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
  const float4 *xyzi = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
  const unsigned long long *alive = w.alive + (int64_t)s * chunks;
  const double q_lo = w.q_ext[2 * s + 0], q_hi = w.q_ext[2 * s + 1];
  const bool tiny_el = (b.bounds[2 * s + 0] - b.bounds[2 * s + 1]) / (double)b.rows < 1e-4;
  const int wpr = b.cols >> 5;
  uint32_t p[kPerThread];
  float4 f[kPerThread];
  bool live[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int j = t0 + k * kPT + threadIdx.x;
    live[k] = j < n && ((alive[j >> 6] >> (j & 63)) & 1ull);
    p[k] = live[k] ? pixs[j] : 0u;
  }
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int j = t0 + k * kPT + threadIdx.x;
    f[k] = make_float4(1.f, 0.f, 0.f, 0.f);
    if (live[k] && !n_virt && j < n_head) f[k] = xyzi[j];
  }
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int j = t0 + k * kPT + threadIdx.x;
    int word = -1;
    uint32_t bit = 0u;
    if (live[k]) {
      double x = (double)f[k].x, y = (double)f[k].y, z = (double)f[k].z;
      if (n_virt || j >= n_head) load_point(b, s, orig_of(w, b, s, n_virt, j), n_head, x, y, z);
      const double ss = x * x + y * y + z * z;
      const int row = pix_row(p[k]), col = pix_col(p[k]);
      atomicMin(&img[row * b.cols + col], depth_key(ss));
      word = row * wpr + (col >> 5);
      bit = 1u << (col & 31);
      if (row == 0 || row == b.rows - 1 || tiny_el) {
        const double q = z / sqrt(ss);
        if (q == q_lo || q == q_hi) {
          const int at = atomicAdd(&w.n_hold[s], 1);
          if (at < kHoldCap) w.hold_pix[(int64_t)s * kHoldCap + at] = (int32_t)p[k];
        }
      }
    }
    or_occ_by_runs(occ, word, bit);
  }
}

__global__ void k_image_validate(r3d_batch_t b, BatchWs w) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < b.B && !w.img_valid[s]) w.img_valid[s] = 1, w.img_dirty[s] = 0;
}

int launch_image_clear(const r3d_batch_t &b, const BatchWs &w, hipStream_t st) {
  const size_t npix = (size_t)b.rows * b.cols;
  int gx = (int)((npix / 2 + kPT * 8 - 1) / (kPT * 8));
  gx = gx < 1 ? 1 : gx;
  hipLaunchKernelGGL(k_image_clear, dim3(gx, b.B), dim3(kPT), 0, st, b, w);
  R3D_LAUNCHED("k_image_clear");
  return R3D_OK;
}

int launch_image_build(const r3d_batch_t &b, const BatchWs &w, hipStream_t st) {
  hipLaunchKernelGGL(k_image_build, dim3(tiles_of(b), b.B), dim3(kPT), 0, st, b, w, chunks_of(b));
  R3D_LAUNCHED("k_image_build");
  return R3D_OK;
}

}  // namespace r3d
