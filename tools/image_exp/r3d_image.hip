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
#include "r3d_insert_core.hpp"   // (built with -I pcl-augmentation_amd/csrc: tools/image_exp/image_build.sh)

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

// Is this living point one that holds an elevation bound?  Only the first / last row can hold one (unless the elevation span
// is tiny): when such a point dies the bounds may move (insertion.py:373 recomputes them for every insert).
__device__ __forceinline__ void note_holder(const r3d_batch_t &b, const BatchWs &w, int s, uint32_t p, double z, double ss,
                                            double q_lo, double q_hi, bool tiny_el) {
  const int row = pix_row(p);
  if (row == 0 || row == b.rows - 1 || tiny_el) {
    const double q = z / sqrt(ss);
    if (q == q_lo || q == q_hi) {
      const int at = atomicAdd(&w.n_hold[s], 1);
      if (at < kHoldCap) w.hold_pix[(int64_t)s * kHoldCap + at] = (int32_t)p;
    }
  }
}

// The LIVING points of the scenes whose image is not valid, min-reduced on the squared depth (insertion.py:118-125) with one
// global atomic per point; the pixels of the points that hold an elevation bound are noted.  A block takes one 2048-point
// tile; the points are numbered as pix / alive number them (virtual order: perm).  band_rows > 0: only the points of the row
// bands k_image_bands has left (band_fall), and only scenes that have such bands -- the image itself was written by
// k_image_bands then (those bands empty).  Measured (round 6, profiles/r06_image_build.md): 47 G points/s on scans in ring
// order, 13 G on shuffled ones -- 0.65 / 2.4 ms per 256 scenes of 120 000 points, more than the chain route's whole insert
// launch: the reason why k_image_bands exists.
__global__ void __launch_bounds__(kPT)
k_image_build(r3d_batch_t b, BatchWs w, int chunks, int band_rows) {
  const int s = blockIdx.y;
  if (w.img_valid[s]) return;
  if (band_rows > 0 && w.n_fall[s] == 0) return;
  const int n = b.n_total[s], n_head = b.n_head[s], n_virt = w.n_virt[s];
  const int t0 = blockIdx.x * kTile;
  if (t0 >= n) return;
  const size_t npix = (size_t)b.rows * b.cols;
  unsigned long long *img = w.img + (size_t)s * npix;
  uint32_t *occ = w.occ + (size_t)s * (npix / 32);
  const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
  const float4 *xyzi = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
  const unsigned long long *alive = w.alive + (int64_t)s * chunks;
  const uint8_t *fall = w.band_fall + (int64_t)s * b.rows;
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
    if (live[k] && band_rows > 0 && !fall[pix_row(p[k]) / band_rows]) live[k] = false;
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
      note_holder(b, w, s, p[k], z, ss, q_lo, q_hi, tiny_el);
    }
    or_occ_by_runs(occ, word, bit);
  }
}

// The same without a global atomic: ONE workgroup per (scene, band of rows) holds the band in LDS, lists the 64-point chunks
// whose box reaches the band (through the super-boxes on large clouds), min-reduces their living points there (ds_min_u64)
// and writes the band out -- image, occupancy words, kill steps zeroed -- with plain coalesced stores: every byte of the
// image is written exactly once, nothing has to be cleared first.  Leans on the chunk boxes, i.e. on the point order, for
// SPEED only: a band whose list exceeds the workgroup's room (a cloud in no order lists every chunk for every band) is
// written empty and flagged, k_image_build then adds its points with global atomics.
constexpr int kBandNT = 1024;
constexpr int kBandListCap = 4096;           // listed chunks a band's workgroup holds
constexpr int kBandSupCap = 2048;            // super-boxes of a scene (64 chunks each) it can list
constexpr int kBandTileBytes = 112 * 1024;
inline int band_rows_of(const r3d_batch_t &b) {
  int r = kBandTileBytes / (b.cols * 8);
  return r < 1 ? 1 : (r > b.rows ? b.rows : r);
}
__global__ void k_image_prepare(r3d_batch_t b, BatchWs w) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= b.B || w.img_valid[s]) return;
  w.n_hold[s] = 0;
  w.n_fall[s] = 0;
}
__global__ void __launch_bounds__(kBandNT)
k_image_bands(r3d_batch_t b, BatchWs w, int chunks, int band_rows) {
  extern __shared__ __align__(16) unsigned char s_band[];
  unsigned long long *tile = reinterpret_cast<unsigned long long *>(s_band);
  uint32_t *list = reinterpret_cast<uint32_t *>(s_band + (size_t)band_rows * b.cols * 8);
  uint16_t *sup = reinterpret_cast<uint16_t *>(list + kBandListCap);
  __shared__ int s_n, s_nsup;
  const int s = blockIdx.y, band = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (w.img_valid[s]) return;
  const int r0 = band * band_rows, r1 = (r0 + band_rows < b.rows ? r0 + band_rows : b.rows) - 1;
  const int npx = (r1 - r0 + 1) * b.cols;
  const int n = b.n_total[s], n_head = b.n_head[s], n_virt = w.n_virt[s];
  const int n_chunks = (n + 63) >> 6;
  for (int i = tid; i < npx; i += kBandNT) tile[i] = R3D_SENT;
  if (tid == 0) s_n = s_nsup = 0;
  __syncthreads();
  // the chunks whose box reaches the band's rows and that have somebody alive
  const unsigned long long *boxes = w.chunk_box + (int64_t)s * chunks;
  const unsigned long long *alive = w.alive + (int64_t)s * chunks;
  const bool supers = supers_on(b, chunks) && ((n_chunks + 63) >> 6) <= kBandSupCap;
  if (supers) {
    const int n_sup_all = (chunks + 63) >> 6, n_sup = (n_chunks + 63) >> 6;
    const int2 *rows2 = reinterpret_cast<const int2 *>(w.super_rows) + (int64_t)s * n_sup_all;
    for (int sp = tid; sp < n_sup; sp += kBandNT) {
      const int2 r = rows2[sp];
      if (r.x <= r1 && r.y >= r0) sup[atomicAdd(&s_nsup, 1)] = (uint16_t)sp;
    }
    __syncthreads();
  }
  const int n_items = supers ? s_nsup << 6 : n_chunks;
  for (int c0 = tid; c0 < n_items; c0 += kBandNT) {
    const int c = supers ? ((int)sup[c0 >> 6] << 6) + (c0 & 63) : c0;
    if (c >= n_chunks) continue;
    const unsigned long long bx = boxes[c];
    const int rmin = (int)(bx & 0xFFFF), rmax = (int)((bx >> 16) & 0xFFFF);
    if (rmin <= r1 && rmax >= r0 && alive[c] != 0ull) {
      const int at = atomicAdd(&s_n, 1);
      if (at < kBandListCap) list[at] = (uint32_t)c;
    }
  }
  __syncthreads();
  int nl = s_n;
  if (nl > kBandListCap) {                                     // left to k_image_build: the band stays empty here
    if (tid == 0) {
      w.band_fall[(int64_t)s * b.rows + band] = 1;
      atomicAdd(&w.n_fall[s], 1);
    }
    nl = 0;
  } else if (tid == 0) {
    w.band_fall[(int64_t)s * b.rows + band] = 0;
  }
  // a wave per listed chunk, two in flight
  const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
  const float4 *xyzi = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
  const double q_lo = w.q_ext[2 * s + 0], q_hi = w.q_ext[2 * s + 1];
  const bool tiny_el = (b.bounds[2 * s + 0] - b.bounds[2 * s + 1]) / (double)b.rows < 1e-4;
  constexpr int kW = kBandNT / 64, kU = 2;
  for (int e0 = wave; e0 < nl; e0 += kU * kW) {
    uint32_t p[kU];
    float4 f[kU];
    int j[kU];
    bool live[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int e = e0 + u * kW;
      const int c = e < nl ? (int)list[e] : 0;
      j[u] = (c << 6) + lane;
      live[u] = e < nl && j[u] < n && ((alive[c] >> lane) & 1ull);
      p[u] = live[u] ? pixs[j[u]] : 0xFFFF0000u;
      f[u] = make_float4(1.f, 0.f, 0.f, 0.f);
      if (live[u] && !n_virt && j[u] < n_head) f[u] = xyzi[j[u]];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int row = pix_row(p[u]), col = pix_col(p[u]);
      if (!live[u] || row < r0 || row > r1) continue;
      double x = (double)f[u].x, y = (double)f[u].y, z = (double)f[u].z;
      if (n_virt || j[u] >= n_head) load_point(b, s, orig_of(w, b, s, n_virt, j[u]), n_head, x, y, z);
      const double ss = x * x + y * y + z * z;
      atomicMin(&tile[(row - r0) * b.cols + col], depth_key(ss));
      note_holder(b, w, s, p[u], z, ss, q_lo, q_hi, tiny_el);
    }
  }
  __syncthreads();
  // the band out: image, occupancy words (a wave's 64 pixels = two words), kill steps
  const size_t npix = (size_t)b.rows * b.cols, g0 = (size_t)r0 * b.cols;
  unsigned long long *img = w.img + (size_t)s * npix + g0;
  uint32_t *occ = w.occ + ((size_t)s * npix + g0) / 32;
  for (int i0 = wave * 64; i0 < npx; i0 += kBandNT) {
    const int i = i0 + lane;
    const unsigned long long v = i < npx ? tile[i] : R3D_SENT;
    if (i < npx) img[i] = v;
    const unsigned long long m = __ballot(v != R3D_SENT);
    if (lane == 0) {
      occ[i0 >> 5] = (uint32_t)m;
      if (i0 + 32 < npx) occ[(i0 >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
  uint4 *ks = reinterpret_cast<uint4 *>(w.kstep + (size_t)s * npix + g0);
  for (int i = tid; i < npx / 8; i += kBandNT) ks[i] = make_uint4(0u, 0u, 0u, 0u);
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
  hipLaunchKernelGGL(k_image_build, dim3(tiles_of(b), b.B), dim3(kPT), 0, st, b, w, chunks_of(b), 0);
  R3D_LAUNCHED("k_image_build");
  return R3D_OK;
}

// The image of every scene that has no valid one: a band per workgroup, then global atomics for the bands that left their
// points (none for clouds in a file order).
int launch_image_bands(const r3d_batch_t &b, const BatchWs &w, hipStream_t st) {
  const int band_rows = band_rows_of(b), n_bands = (b.rows + band_rows - 1) / band_rows;
  const size_t lds = (size_t)band_rows * b.cols * 8 + kBandListCap * 4 + kBandSupCap * 2;
  if (lds > 160 * 1024 - 64) return fail(R3D_E_ARG, "image route: a row of the range image exceeds a workgroup's LDS");
  R3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_image_bands), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_image_prepare, dim3((b.B + 255) / 256), dim3(256), 0, st, b, w);
  hipLaunchKernelGGL(k_image_bands, dim3(n_bands, b.B), dim3(kBandNT), lds, st, b, w, chunks_of(b), band_rows);
  hipLaunchKernelGGL(k_image_build, dim3(tiles_of(b), b.B), dim3(kPT), 0, st, b, w, chunks_of(b), band_rows);
  R3D_LAUNCHED("k_image_bands");
  return R3D_OK;
}

}  // namespace r3d
