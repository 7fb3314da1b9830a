// Rich-map rasterisation for gfx950.
//
// Reference: the __main__ block of semantic_segmentation/rich_map/drivable_area_map.py:122-206.  Pass
// one takes every frame of a sequence to world coordinates (`t_matrix @ points.T`, :139-141) and
// tracks the extremes of x and y (:143-150); pass two walks the points of every frame in order and
// writes the cell under each placement-surface point (:172-200): road 1 and sidewalk 2 overwrite
// each other, parking 3 is never overwritten.  So a cell ends as 3 if any parking point fell on
// it, otherwise as the code of the LAST road / sidewalk point in (frame, point) order.  Here every
// point does one 64-bit atomicMax per cell with the key (is-parking, frame number, point index,
// code): the maximum is exactly that last writer.
#include "r3d_device.hpp"
#include "r3d_host.hpp"

namespace {
using namespace r3d;

struct Pose {
  double t[16];
};
struct SurfaceLabels {      // config['insertion']['placement_labels'][1..3]
  int32_t label[3][8];
  int32_t n[3];
};

// t_matrix @ [x y z 1]^T the way the BLAS product accumulates (a0*b0, fma, fma, fma), then the
// division by the homogeneous coordinate (:140-141).
__device__ __forceinline__ void to_world(const Pose &p, float4 v, double &x, double &y) {
  double px = (double)v.x, py = (double)v.y, pz = (double)v.z;
  double r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = fma(p.t[i * 4 + 3], 1.0, fma(p.t[i * 4 + 2], pz, fma(p.t[i * 4 + 1], py, p.t[i * 4 + 0] * px)));
  x = r[0] / r[3];
  y = r[1] / r[3];
}

__global__ __launch_bounds__(256) void k_map_bounds(const float4 *xyzi, int64_t n, Pose pose,
                                                   unsigned long long *minmax) {
  __shared__ unsigned long long s_v[4][4];
  unsigned long long lo_x = ~0ull, hi_x = 0ull, lo_y = ~0ull, hi_y = 0ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double x, y;
    to_world(pose, xyzi[i], x, y);
    unsigned long long kx = ordered_key(x), ky = ordered_key(y);
    lo_x = kx < lo_x ? kx : lo_x;
    hi_x = kx > hi_x ? kx : hi_x;
    lo_y = ky < lo_y ? ky : lo_y;
    hi_y = ky > hi_y ? ky : hi_y;
  }
  lo_x = wave_min_u64(lo_x);
  hi_x = wave_max_u64(hi_x);
  lo_y = wave_min_u64(lo_y);
  hi_y = wave_max_u64(hi_y);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_v[0][wave] = lo_x;
    s_v[1][wave] = hi_x;
    s_v[2][wave] = lo_y;
    s_v[3][wave] = hi_y;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long v = s_v[threadIdx.x][0];
    for (int w = 1; w < 4; ++w) {
      unsigned long long u = s_v[threadIdx.x][w];
      v = (threadIdx.x & 1) ? (u > v ? u : v) : (u < v ? u : v);
    }
    if (threadIdx.x & 1) atomicMax(&minmax[threadIdx.x], v);
    else atomicMin(&minmax[threadIdx.x], v);
  }
}

__global__ __launch_bounds__(256) void k_map_splat(const float4 *xyzi, const uint32_t *label, int64_t n, Pose pose,
                                                  SurfaceLabels sl, double min_x, double min_y, int size_x,
                                                  int size_y, unsigned long long frame_no, unsigned long long *keys,
                                                  int32_t *status) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int lab = (int)(label[i] & 0xFFFFu);                      // tools/datasets.py:54 semantic part
    int code = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < sl.n[c]; ++j)
        if (lab == sl.label[c][j] && code == 0) code = c + 1;
    if (!code) continue;                                       // :175-176
    double x, y;
    to_world(pose, xyzi[i], x, y);
    double px = x - min_x, py = y - min_y;                     // :178-179
    if (!(px >= 0.0 && py >= 0.0) || !(px < (double)size_x && py < (double)size_y)) {
      atomicOr(status, 1);                                     // the assert of :180 / an index error
      continue;
    }
    unsigned long long key = (code == 3 ? 1ull << 63 : 0ull) | (frame_no << 34) | ((unsigned long long)i << 2) |
                             (unsigned long long)code;
    atomicMax(&keys[(size_t)(int)px * size_y + (int)py], key);
  }
}

__global__ void k_map_finish(const unsigned long long *keys, int64_t cells, double *map64, uint8_t *map8) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  int code = (int)(keys[i] & 3ull);
  if (map64) map64[i] = (double)code;
  if (map8) map8[i] = (uint8_t)code;
}

// ---- object-detection flavour: object_detection/rich_map/single_drivable_area_map.py:113-194, one frame ----
// :129-139 the cells under the frame's Road points (frame coordinates, int() truncation).
__global__ __launch_bounds__(256) void k_od_splat(const float4 *xyzi, const uint32_t *label, int64_t n, int road_label,
                                                 int min_x, int min_y, int size_x, int size_y, uint8_t *raster) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if ((int)(label[i] & 0xFFFFu) != road_label) continue;                  // tools/datasets.py:60-61, :130
    float4 v = xyzi[i];
    int px = (int)((double)v.x - (double)min_x), py = (int)((double)v.y - (double)min_y);   // :133-134, :138
    if (px >= 0 && py >= 0 && px < size_x && py < size_y) raster[(size_t)px * size_y + py] = 1;
  }
}

// Binary dilation (erode = 0) or erosion (erode = 1) with disk(radius) -- the cells (dr, dc) with
// dr*dr + dc*dc <= radius*radius -- windows clipped at the borders (what scikit-image's reflecting grey
// morphology amounts to for a symmetric convex footprint).  :145-151 closing = erosion of the dilation
// with disk(4); :182-188 dilation with disk(2).
__global__ __launch_bounds__(256) void k_od_morph(const uint8_t *src, uint8_t *dst, int size_x, int size_y, int radius,
                                                 int erode) {
  int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= size_x * size_y) return;
  int r = cell / size_y, c = cell - r * size_y;
  int acc = erode ? 1 : 0;
  for (int dr = -radius; dr <= radius; ++dr)
    for (int dc = -radius; dc <= radius; ++dc) {
      if (dr * dr + dc * dc > radius * radius) continue;
      int rr = r + dr, cc = c + dc;
      if (rr < 0 || rr >= size_x || cc < 0 || cc >= size_y) continue;
      int v = src[(size_t)rr * size_y + cc] != 0;
      acc = erode ? (acc & v) : (acc | v);
    }
  dst[cell] = (uint8_t)acc;
}

// :160-178 cells that are not road but touch a road cell (8-neighbourhood, clipped).
__global__ __launch_bounds__(256) void k_od_ring(const uint8_t *road, uint8_t *ring, int size_x, int size_y) {
  int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= size_x * size_y) return;
  int r = cell / size_y, c = cell - r * size_y;
  int near = 0;
  if (road[cell] == 0)
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        if (rr >= 0 && rr < size_x && cc >= 0 && cc < size_y && road[(size_t)rr * size_y + cc] == 1) near = 1;
      }
  ring[cell] = (uint8_t)near;
}

int fill_pose(const double *pose16, Pose &p) {
  if (!pose16) return fail(R3D_E_ARG, "rich map: null pose");
  for (int i = 0; i < 16; ++i) p.t[i] = pose16[i];
  return R3D_OK;
}
}  // namespace

extern "C" int r3d_map_bounds(const float *xyzi, int64_t n, const double *pose16, uint64_t *minmax, void *stream) {
  if (!xyzi || !minmax || n < 0) return fail(R3D_E_ARG, "map_bounds: bad argument");
  Pose p;
  int rc = fill_pose(pose16, p);
  if (rc != R3D_OK || n == 0) return rc;
  hipLaunchKernelGGL(k_map_bounds, dim3(blocks_for(n, 256 * 8, 2048)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(xyzi), n, p, reinterpret_cast<unsigned long long *>(minmax));
  R3D_LAUNCHED("k_map_bounds");
  return R3D_OK;
}

extern "C" int r3d_map_splat(const float *xyzi, const uint32_t *label, int64_t n, const double *pose16,
                             const int32_t *labels_road, int32_t n_road, const int32_t *labels_sidewalk,
                             int32_t n_sidewalk, const int32_t *labels_parking, int32_t n_parking, double min_x,
                             double min_y, int32_t size_x, int32_t size_y, int64_t frame_no, uint64_t *keys,
                             int32_t *status, void *stream) {
  if (!xyzi || !label || !keys || !status || n < 0 || size_x <= 0 || size_y <= 0)
    return fail(R3D_E_ARG, "map_splat: bad argument");
  if (n_road < 0 || n_road > 8 || n_sidewalk < 0 || n_sidewalk > 8 || n_parking < 0 || n_parking > 8)
    return fail(R3D_E_ARG, "map_splat: at most 8 labels per surface");
  if (frame_no < 0 || frame_no >= (1ll << 29) || n >= (1ll << 32))
    return fail(R3D_E_ARG, "map_splat: frame number or point count too large for the cell keys");
  Pose p;
  int rc = fill_pose(pose16, p);
  if (rc != R3D_OK || n == 0) return rc;
  SurfaceLabels sl{};
  const int32_t *src[3] = {labels_road, labels_sidewalk, labels_parking};
  const int32_t cnt[3] = {n_road, n_sidewalk, n_parking};
  for (int c = 0; c < 3; ++c) {
    sl.n[c] = cnt[c];
    for (int j = 0; j < cnt[c]; ++j) sl.label[c][j] = src[c][j];
  }
  hipLaunchKernelGGL(k_map_splat, dim3(blocks_for(n, 256 * 4, 4096)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(xyzi), label, n, p, sl, min_x, min_y, (int)size_x, (int)size_y,
                     (unsigned long long)frame_no, reinterpret_cast<unsigned long long *>(keys), status);
  R3D_LAUNCHED("k_map_splat");
  return R3D_OK;
}

extern "C" int r3d_map_finish(const uint64_t *keys, int64_t cells, double *map64, uint8_t *map8, void *stream) {
  if (!keys || cells <= 0 || (!map64 && !map8)) return fail(R3D_E_ARG, "map_finish: bad argument");
  hipLaunchKernelGGL(k_map_finish, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long *>(keys), cells, map64, map8);
  R3D_LAUNCHED("k_map_finish");
  return R3D_OK;
}

extern "C" int r3d_od_maps(const float *xyzi, const uint32_t *label, int64_t n, int32_t road_label, int32_t min_x,
                           int32_t min_y, int32_t size_x, int32_t size_y, uint8_t *road_map, uint8_t *pedestrian_map,
                           uint8_t *scratch, void *stream) {
  if (!xyzi || !label || !road_map || !pedestrian_map || !scratch || n < 0 || size_x <= 0 || size_y <= 0 ||
      (int64_t)size_x * size_y > (1ll << 30))
    return fail(R3D_E_ARG, "od_maps: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cells = size_x * size_y;
  uint8_t *a = scratch, *b = scratch + cells;
  R3D_HIP(hipMemsetAsync(a, 0, (size_t)cells, st));
  if (n > 0)
    hipLaunchKernelGGL(k_od_splat, dim3(blocks_for(n, 256 * 4, 4096)), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(xyzi), label, n, (int)road_label, (int)min_x, (int)min_y, (int)size_x,
                       (int)size_y, a);
  const dim3 grid((cells + 255) / 256), block(256);
  hipLaunchKernelGGL(k_od_morph, grid, block, 0, st, a, b, (int)size_x, (int)size_y, 4, 0);
  hipLaunchKernelGGL(k_od_morph, grid, block, 0, st, b, road_map, (int)size_x, (int)size_y, 4, 1);
  hipLaunchKernelGGL(k_od_ring, grid, block, 0, st, road_map, a, (int)size_x, (int)size_y);
  hipLaunchKernelGGL(k_od_morph, grid, block, 0, st, a, pedestrian_map, (int)size_x, (int)size_y, 2, 0);
  R3D_LAUNCHED("od_maps kernels");
  return R3D_OK;
}
