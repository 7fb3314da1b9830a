// Level 1: the reference's hot-path functions one for one, on its float64 N x 9 layout.
// Each kernel cites the reference lines it implements (paths relative to the reference root,
// SS = semantic_segmentation/).  Built for gfx950 only, -ffp-contract=off.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "r3d_device.hpp"
#include "r3d_host.hpp"

namespace r3d {

std::string &last_error_ref() {
  static thread_local std::string s;
  return s;
}

constexpr int kThreads = 256;
constexpr unsigned kNoKey = 0xFFFFFFFFu;   // sorts after every pixel id (ids are non-negative 32-bit ints)

// ---------------------------------------------------------------------------------------------
// a1  add_space_for_spherical                                  SS Real3DAug/insertion.py:54-64
// ---------------------------------------------------------------------------------------------
__global__ void k_add_space(const double *__restrict__ pcl5, int64_t n, double *__restrict__ pcl9) {
  int64_t total = n * 9;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / 9;
    int c = (int)(e - i * 9);
    double v = -1.0;                                   // :61
    if (c < 3) v = pcl5[i * 5 + c];                    // :62
    else if (c == 6) v = pcl5[i * 5 + 3];              // :63
    else if (c == 7) v = pcl5[i * 5 + 4];
    pcl9[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// a2  fill_spherical                                           SS Real3DAug/insertion.py:67-81
// ---------------------------------------------------------------------------------------------
__global__ void k_init_bounds(unsigned long long *bounds_bits) {
  bounds_bits[0] = 0ull;                               // running max of elevation (>= 0)
  bounds_bits[1] = 0x7FF0000000000000ull;              // running min: +inf
}

__global__ void __launch_bounds__(kThreads)
k_fill_spherical(double *__restrict__ pcl9, int64_t n, unsigned long long *bounds_bits, int32_t *status) {
  __shared__ unsigned long long s_min[kThreads / 64], s_max[kThreads / 64];
  unsigned long long lmin = 0x7FF0000000000000ull, lmax = 0ull;
  int bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double *p = pcl9 + i * 9;
    Sph s = spherical(p[0], p[1], p[2]);               // :74-76
    p[3] = s.r;
    p[4] = s.az;
    p[5] = s.el;
    if (!(s.el >= 0.0) || !isfinite(s.az) || !isfinite(s.r)) {
      bad = 1;                                         // NaN / Inf / r == 0
    } else {
      unsigned long long k = depth_key(s.el);          // el in [0, pi]: bit order == value order
      lmin = k < lmin ? k : lmin;
      lmax = k > lmax ? k : lmax;
    }
  }
  lmin = wave_min_u64(lmin);
  lmax = wave_max_u64(lmax);
  bad = wave_or_i32(bad);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_min[wave] = lmin;
    s_max[wave] = lmax;
    if (bad) atomicOr(status, R3D_S_NONFINITE);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) {
      lmin = s_min[w] < lmin ? s_min[w] : lmin;
      lmax = s_max[w] > lmax ? s_max[w] : lmax;
    }
    atomicMax(&bounds_bits[0], lmax);                  // :79
    atomicMin(&bounds_bits[1], lmin);                  // :78
  }
}

// ---------------------------------------------------------------------------------------------
// a3  geometrical_front_view                                   SS Real3DAug/insertion.py:84-129
// ---------------------------------------------------------------------------------------------
__global__ void k_fill_u64(unsigned long long *p, int64_t n, unsigned long long v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void __launch_bounds__(kThreads)
k_front_view(double *__restrict__ pcl9, int64_t n, int rows, int cols, int id_cols, double max_el, double min_el,
             int sample, unsigned long long *__restrict__ grid, int32_t *status) {
  Binning b = make_binning(max_el, min_el, rows, cols);
  int flags = 0;
  const int lane = threadIdx.x & 63;
  // whole waves walk the cloud together so that the shuffles below always see all 64 lanes
  const int64_t n_round = (n + 63) & ~63ll;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_round;
       i += (int64_t)gridDim.x * blockDim.x) {
    long long cell = -1 - lane;                        // no pixel: unique, never equal to a neighbour's
    unsigned long long key = R3D_SENT;
    if (i < n) {
      double *p = pcl9 + i * 9;
      int row, col;
      int ok = bin_point(b, p[4], p[5], row, col);     // :104-105
      if (!(ok & 1)) {
        if (!sample) flags |= R3D_S_ROW_RANGE;         // assert :110; sample: skipped, :107-108
      } else if (!(ok & 2)) {
        flags |= R3D_S_COL_RANGE;                      // assert :112
      } else {
        p[8] = (double)(row * id_cols + col);          // :116 / :127 (the global NUMCOLUMN, not the argument)
        cell = (long long)row * cols + col;
        key = depth_key(p[3]);
      }
    }
    // LiDAR files are ring-ordered: neighbouring lanes often hit the same pixel.  A segmented
    // min over runs of equal pixels (wave shuffles) leaves one atomic per run (:118-125).
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long k2 = __shfl_down(key, o, 64);
      long long c2 = __shfl_down(cell, o, 64);
      if (lane + o < 64 && c2 == cell) key = k2 < key ? k2 : key;
    }
    long long prev = __shfl_up(cell, 1, 64);
    if (cell >= 0 && (lane == 0 || prev != cell)) atomicMin(&grid[cell], key);
  }
  flags = wave_or_i32(flags);
  if ((threadIdx.x & 63) == 0 && flags) atomicOr(status, flags);
}

__global__ void k_grid_export(const unsigned long long *__restrict__ grid, int64_t npix,
                              double *__restrict__ train, double *__restrict__ label) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix;
       p += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long k = grid[p];
    train[p] = k == R3D_SENT ? R3D_EMPTY_DEPTH : key_depth(k);      // :99
    label[p] = k == R3D_SENT ? -1.0 : 1.0;                          // :98, :123
  }
}

// ---------------------------------------------------------------------------------------------
// a4 / a5  class_closing, smooth_out                      SS Real3DAug/tools/closing.py:9-62
// One 16 x 64 pixel tile per 256-thread block; the grey image and its dilation are staged in LDS
// with their halos (4 rows / 2 columns and 2 rows / 1 column).
// ---------------------------------------------------------------------------------------------
constexpr int kTR = 16, kTC = 64;

__device__ __forceinline__ unsigned char as_ubyte(double label) {
  double v = label < 0.0 ? 0.0 : (label > 1.0 ? 1.0 : label);       // closing.py:17 clip
  return (unsigned char)rint(v * 255.0);                            // closing.py:19 img_as_ubyte
}

template <bool FILL>
__global__ void __launch_bounds__(kThreads)
k_closing(const double *__restrict__ train, const double *__restrict__ label, int rows, int cols,
          unsigned char *__restrict__ closed_out, double *__restrict__ train_out,
          double *__restrict__ label_out) {
  __shared__ unsigned char s_v[kTR + 8][kTC + 4];
  __shared__ unsigned char s_d[kTR + 4][kTC + 2];
  const int r0 = blockIdx.y * kTR, c0 = blockIdx.x * kTC;
  for (int e = threadIdx.x; e < (kTR + 8) * (kTC + 4); e += kThreads) {
    int lr = e / (kTC + 4), lc = e % (kTC + 4);
    int r = r0 - 4 + lr, c = c0 - 2 + lc;
    unsigned char v = 0;                               // outside the image: neutral for the max
    if (r >= 0 && r < rows && c >= 0 && c < cols) v = as_ubyte(label[(int64_t)r * cols + c]);
    s_v[lr][lc] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < (kTR + 4) * (kTC + 2); e += kThreads) {
    int lr = e / (kTC + 2), lc = e % (kTC + 2);
    int r = r0 - 2 + lr, c = c0 - 1 + lc;
    unsigned char d = 255;                             // outside the image: neutral for the min
    if (r >= 0 && r < rows && c >= 0 && c < cols) {
      d = 0;
      for (int dr = 0; dr < 5; ++dr)
        for (int dc = 0; dc < 3; ++dc) {
          unsigned char v = s_v[lr + dr][lc + dc];
          d = v > d ? v : d;
        }
    }
    s_d[lr][lc] = d;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kTR * kTC; e += kThreads) {
    int lr = e / kTC, lc = e % kTC;
    int r = r0 + lr, c = c0 + lc;
    if (r >= rows || c >= cols) continue;
    unsigned char cl = 255;
    for (int dr = 0; dr < 5; ++dr)
      for (int dc = 0; dc < 3; ++dc) {
        unsigned char d = s_d[lr + dr][lc + dc];
        cl = d < cl ? d : cl;
      }
    int64_t p = (int64_t)r * cols + c;
    if (!FILL) {
      closed_out[p] = cl;                              // closing.py:21
      continue;
    }
    double lab = label[p], tr = train[p];
    if (!((cl == 255 && lab == 1.0) || cl == 0)) {     // closing.py:40-42
      int neighbors = 0;
      double sum = 0.0;
      for (int dr = -2; dr <= 2; ++dr)                 // closing.py:46
        for (int dc = -1; dc <= 1; ++dc) {             // closing.py:47
          int rr = r + dr, cc = c + dc;
          if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
          if (label[(int64_t)rr * cols + cc] == 1.0) { // closing.py:48-49
            ++neighbors;
            sum += train[(int64_t)rr * cols + cc];     // closing.py:51
          }
        }
      if (neighbors != 0) tr = sum / (double)neighbors;   // closing.py:57
      lab = 1.0;                                          // closing.py:55, :58
    }
    train_out[p] = tr;
    label_out[p] = lab;
  }
}

// ---------------------------------------------------------------------------------------------
// a6-a8  visibility mask, cull, select                     SS Real3DAug/insertion.py:463-482
// ---------------------------------------------------------------------------------------------
__global__ void k_vis_mask(const double *__restrict__ scene_train, const double *__restrict__ sample_train,
                           int64_t npix, uint32_t *__restrict__ vis_words) {
  int64_t base = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) & ~63ll;
  int lane = threadIdx.x & 63;
  for (; base < npix; base += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = base + lane;
    bool v = p < npix && sample_train[p] < scene_train[p];           // :467
    unsigned long long m = __ballot(v);
    if (lane == 0) {
      vis_words[base / 32] = (uint32_t)m;
      vis_words[base / 32 + 1] = (uint32_t)(m >> 32);
    }
  }
}

__device__ __forceinline__ bool pid_visible(double pid, int rows, int cols, int id_cols, const uint32_t *vis_words) {
  if (!(pid >= 0.0)) return false;                     // -1: never binned (:107-108)
  int id = (int)pid;
  int row = id / id_cols, col = id - row * id_cols;    // :470
  if (row >= rows || col >= cols) return false;
  int p = row * cols + col;
  return (vis_words[p >> 5] >> (p & 31)) & 1u;
}

// keys[i] = (pixel id of a hit row | kNoKey) << 32 | i; per-tile counts of kept rows.
constexpr int kTile = 2048;

__global__ void __launch_bounds__(kThreads)
k_mark(const double *__restrict__ pcl9, int64_t n, int rows, int cols, int id_cols, const uint32_t *__restrict__ vis_words,
       unsigned long long *__restrict__ keys, int32_t *__restrict__ tile_keep, int64_t *hit_count) {
  __shared__ int s_cnt[kThreads / 64];
  int64_t t0 = (int64_t)blockIdx.x * kTile;
  int keep = 0;
  for (int k = threadIdx.x; k < kTile; k += kThreads) {
    int64_t i = t0 + k;
    if (i >= n) break;
    double pid = pcl9[i * 9 + 8];
    bool hit = pid_visible(pid, rows, cols, id_cols, vis_words);
    // unique 64-bit key (pixel id, row index): the sort order does not lean on stability
    keys[i] = ((unsigned long long)(hit ? (uint32_t)(int)pid : kNoKey) << 32) | (uint32_t)i;
    keep += hit ? 0 : 1;
  }
  keep = wave_sum_i32(keep);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = keep;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kThreads / 64; ++w) tot += s_cnt[w];
    if (tile_keep) tile_keep[blockIdx.x] = tot;
    int64_t here = (n - t0 < kTile ? n - t0 : kTile);
    atomicAdd((unsigned long long *)hit_count, (unsigned long long)(here - tot));
  }
}

// Exclusive scan of the tile counts by one block; writes the grand total to *total.
__global__ void __launch_bounds__(1024)
k_scan_tiles(int32_t *__restrict__ tile_counts, int n_tiles, int64_t *total) {
  __shared__ int sm[1024 / 64 + 1];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_tiles; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < n_tiles ? tile_counts[i] : 0;
    int tot;
    int ex = block_escan_i32(v, sm, tot);
    int carry = s_carry;
    if (i < n_tiles) tile_counts[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}

// scene_out = rows that are not hit, original order (:472-473).
__global__ void __launch_bounds__(kThreads)
k_scatter_keep(const double *__restrict__ pcl9, int64_t n, const unsigned long long *__restrict__ keys,
               const int32_t *__restrict__ tile_offset, double *__restrict__ out9) {
  __shared__ int sm[kThreads / 64 + 1];
  int64_t t0 = (int64_t)blockIdx.x * kTile;
  int base = tile_offset[blockIdx.x];
  for (int k0 = 0; k0 < kTile; k0 += kThreads) {
    int64_t i = t0 + k0 + threadIdx.x;
    int keep = (i < n && (uint32_t)(keys[i] >> 32) == kNoKey) ? 1 : 0;
    int tot;
    int ex = block_escan_i32(keep, sm, tot);
    if (keep) {
      const double *s = pcl9 + i * 9;
      double *d = out9 + (int64_t)(base + ex) * 9;
#pragma unroll
      for (int c = 0; c < 9; ++c) d[c] = s[c];
    }
    base += tot;
  }
}

// rows of src9 picked through the sorted index list, first *count entries.
__global__ void k_gather_rows(const double *__restrict__ src9, const unsigned long long *__restrict__ idx,
                              const int64_t *count, int64_t cap, double *__restrict__ dst9) {
  int64_t m = *count < cap ? *count : cap;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < m * 9;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = e / 9;
    int c = (int)(e - j * 9);
    dst9[e] = src9[(int64_t)(uint32_t)idx[j] * 9 + c];
  }
}

// ---------------------------------------------------------------------------------------------
// a9  remove_space_for_spherical + the save_data casts   SS tools/datasets.py:72-106, OD :76-109
// ---------------------------------------------------------------------------------------------
__global__ void k_remove_space(const double *__restrict__ pcl9, int64_t n, float *__restrict__ xyzi,
                               uint32_t *__restrict__ label, float *__restrict__ check, int check_cols) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double *p = pcl9 + i * 9;
    float x = (float)p[0], y = (float)p[1], z = (float)p[2], in = (float)p[6];   // astype(float32)
    if (xyzi) {
      xyzi[i * 4 + 0] = x;
      xyzi[i * 4 + 1] = y;
      xyzi[i * 4 + 2] = z;
      xyzi[i * 4 + 3] = in;
    }
    if (label) label[i] = (uint32_t)(int64_t)p[7];                               // astype(uint32)
    if (check) {
      float *c = check + i * check_cols;
      c[0] = x;
      c[1] = y;
      c[2] = z;
      c[3] = in;
      if (check_cols == 5) c[4] = (float)p[7];
    }
  }
}

struct MergeWs {
  uint32_t *vis_words;
  unsigned long long *skey, *skey2, *mkey, *mkey2;
  int32_t *tiles;
  void *sort_tmp;
  size_t sort_bytes, total;
};

static hipError_t sort_query(size_t &bytes, int64_t n) {
  bytes = 0;
  unsigned long long *k = nullptr;
  return rocprim::radix_sort_keys(nullptr, bytes, k, k, (size_t)(n < 1 ? 1 : n), 0, 64);
}

static int carve_merge(MergeWs &w, void *ws, int64_t n, int64_t m, int rows, int cols) {
  size_t sb_n = 0, sb_m = 0;
  if (sort_query(sb_n, n) != hipSuccess || sort_query(sb_m, m) != hipSuccess)
    return fail(R3D_E_HIP, "rocprim radix_sort_pairs size query failed");
  w.sort_bytes = sb_n > sb_m ? sb_n : sb_m;
  Carver c(ws);
  int64_t npix = (int64_t)rows * cols;
  w.vis_words = c.take<uint32_t>((size_t)((npix + 63) / 64 * 2));
  w.skey = c.take<unsigned long long>((size_t)n);
  w.skey2 = c.take<unsigned long long>((size_t)n);
  w.mkey = c.take<unsigned long long>((size_t)m);
  w.mkey2 = c.take<unsigned long long>((size_t)m);
  w.tiles = c.take<int32_t>((size_t)((n + kTile - 1) / kTile + 1));
  w.sort_tmp = c.take<char>(w.sort_bytes);
  w.total = c.off;
  return R3D_OK;
}

}  // namespace r3d

using namespace r3d;

extern "C" {

int r3d_version(void) { return R3D_VERSION; }

#ifndef R3D_SRC_HASH
#define R3D_SRC_HASH "unknown"
#endif
const char *r3d_build_info(void) { return "sources " R3D_SRC_HASH; }
const char *r3d_last_error(void) { return last_error_ref().c_str(); }

int r3d_add_space_for_spherical(const double *pcl5, int64_t n, double *pcl9, void *stream) {
  if (n < 0 || (n > 0 && (!pcl5 || !pcl9))) return fail(R3D_E_ARG, "add_space: bad argument");
  if (n == 0) return R3D_OK;
  hipLaunchKernelGGL(k_add_space, dim3(blocks_for(n * 9, kThreads, 4096)), dim3(kThreads), 0,
                     (hipStream_t)stream, pcl5, n, pcl9);
  R3D_LAUNCHED("k_add_space");
  return R3D_OK;
}

int r3d_fill_spherical(double *pcl9, int64_t n, double *bounds, int32_t *status, void *stream) {
  if (n <= 0 || !pcl9 || !bounds || !status)
    return fail(R3D_E_ARG, "fill_spherical: empty cloud or null pointer (reference raises, insertion.py:78)");
  auto *bits = reinterpret_cast<unsigned long long *>(bounds);
  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(1), 0, (hipStream_t)stream, bits);
  hipLaunchKernelGGL(k_fill_spherical, dim3(blocks_for(n, kThreads, 2048)), dim3(kThreads), 0,
                     (hipStream_t)stream, pcl9, n, bits, status);
  R3D_LAUNCHED("k_fill_spherical");
  return R3D_OK;
}

size_t r3d_front_view_workspace_bytes(int32_t num_row, int32_t num_column) {
  if (num_row <= 0 || num_column <= 0) return 0;
  return align_up((size_t)num_row * num_column * sizeof(unsigned long long));
}

int r3d_geometrical_front_view(double *pcl9, int64_t n, int32_t num_row, int32_t num_column,
                               double max_el, double min_el, int32_t sample, double *train,
                               double *label, void *workspace, size_t workspace_bytes,
                               int32_t *status, void *stream) {
  return r3d_geometrical_front_view_grid(pcl9, n, num_row, num_column, R3D_NUMCOLUMN, max_el, min_el, sample, train, label,
                                         workspace, workspace_bytes, status, stream);
}

int r3d_geometrical_front_view_grid(double *pcl9, int64_t n, int32_t num_row, int32_t num_column, int32_t id_columns,
                                    double max_el, double min_el, int32_t sample, double *train,
                                    double *label, void *workspace, size_t workspace_bytes,
                                    int32_t *status, void *stream) {
  if (n < 0 || num_row <= 0 || num_column <= 0 || id_columns <= 0 || (int64_t)num_row * id_columns > 0x7FFFFFFFll || !train ||
      !label || !status || !workspace || (n > 0 && !pcl9))
    return fail(R3D_E_ARG, "front_view: bad argument");
  if (workspace_bytes < r3d_front_view_workspace_bytes(num_row, num_column))
    return fail(R3D_E_WORKSPACE, "front_view: workspace too small");
  auto *grid = static_cast<unsigned long long *>(workspace);
  int64_t npix = (int64_t)num_row * num_column;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_u64, dim3(blocks_for(npix, kThreads, 1024)), dim3(kThreads), 0, st, grid,
                     npix, R3D_SENT);
  if (n > 0)
    hipLaunchKernelGGL(k_front_view, dim3(blocks_for(n, kThreads, 2048)), dim3(kThreads), 0, st, pcl9,
                       n, num_row, num_column, id_columns, max_el, min_el, sample, grid, status);
  hipLaunchKernelGGL(k_grid_export, dim3(blocks_for(npix, kThreads, 1024)), dim3(kThreads), 0, st,
                     grid, npix, train, label);
  R3D_LAUNCHED("front_view kernels");
  return R3D_OK;
}

int r3d_class_closing(const double *label, int32_t rows, int32_t cols, uint8_t *closed, void *stream) {
  if (rows <= 0 || cols <= 0 || !label || !closed) return fail(R3D_E_ARG, "class_closing: bad argument");
  dim3 g((cols + kTC - 1) / kTC, (rows + kTR - 1) / kTR);
  hipLaunchKernelGGL(k_closing<false>, g, dim3(kThreads), 0, (hipStream_t)stream, nullptr, label, rows,
                     cols, closed, nullptr, nullptr);
  R3D_LAUNCHED("k_closing");
  return R3D_OK;
}

int r3d_smooth_out(const double *train, const double *label, int32_t rows, int32_t cols,
                   double *train_out, double *label_out, void *stream) {
  if (rows <= 0 || cols <= 0 || !train || !label || !train_out || !label_out)
    return fail(R3D_E_ARG, "smooth_out: bad argument");
  if (train == train_out || label == label_out)
    return fail(R3D_E_ARG, "smooth_out: outputs must not alias inputs (closing.py:34-35 copies)");
  dim3 g((cols + kTC - 1) / kTC, (rows + kTR - 1) / kTR);
  hipLaunchKernelGGL(k_closing<true>, g, dim3(kThreads), 0, (hipStream_t)stream, train, label, rows,
                     cols, nullptr, train_out, label_out);
  R3D_LAUNCHED("k_closing<fill>");
  return R3D_OK;
}

size_t r3d_occlusion_merge_workspace_bytes(int64_t n, int64_t m, int32_t rows, int32_t cols) {
  if (n < 0 || m < 0 || rows <= 0 || cols <= 0) return 0;
  MergeWs w;
  // carve against a null base: only the offsets matter
  if (carve_merge(w, nullptr, n, m, rows, cols) != R3D_OK) return 0;
  return w.total;
}

int r3d_occlusion_merge(const double *scene9, int64_t n, const double *sample9, int64_t m,
                        const double *scene_train, const double *sample_train, int32_t rows,
                        int32_t cols, double *scene_out9, double *visible9, double *covered9,
                        int64_t *counts, void *workspace, size_t workspace_bytes, void *stream) {
  return r3d_occlusion_merge_grid(scene9, n, sample9, m, scene_train, sample_train, rows, cols, R3D_NUMCOLUMN, scene_out9,
                                  visible9, covered9, counts, workspace, workspace_bytes, stream);
}

int r3d_occlusion_merge_grid(const double *scene9, int64_t n, const double *sample9, int64_t m,
                             const double *scene_train, const double *sample_train, int32_t rows,
                             int32_t cols, int32_t id_columns, double *scene_out9, double *visible9, double *covered9,
                             int64_t *counts, void *workspace, size_t workspace_bytes, void *stream) {
  if (n < 0 || m < 0 || rows <= 0 || cols <= 0 || cols > id_columns || (int64_t)rows * id_columns > 0x7FFFFFFFll ||
      !scene_train || !sample_train || !counts || !workspace || (n > 0 && (!scene9 || !scene_out9 || !covered9)) ||
      (m > 0 && (!sample9 || !visible9)))
    return fail(R3D_E_ARG, "occlusion_merge: bad argument (cols must be <= the id stride, R3D_NUMCOLUMN unless given)");
  // the sort keys: (pixel id << 32) | row index; bits of the largest pixel id and of kNoKey (it sorts last)
  int key_end = 33;
  while (key_end < 64 && ((int64_t)rows * id_columns) >> (key_end - 32)) ++key_end;
  ++key_end;                                             // one more: kNoKey's leading ones beat every id
  if (key_end > 64) key_end = 64;
  MergeWs w;
  int rc = carve_merge(w, workspace, n, m, rows, cols);
  if (rc != R3D_OK) return rc;
  if (workspace_bytes < w.total) return fail(R3D_E_WORKSPACE, "occlusion_merge: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int64_t npix = (int64_t)rows * cols;
  R3D_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), st));
  hipLaunchKernelGGL(k_vis_mask, dim3(blocks_for(npix, kThreads, 1024)), dim3(kThreads), 0, st,
                     scene_train, sample_train, npix, w.vis_words);
  int n_tiles = (int)((n + kTile - 1) / kTile);
  if (n > 0) {
    // counts[2] temporarily accumulates the scene hit count
    hipLaunchKernelGGL(k_mark, dim3(n_tiles), dim3(kThreads), 0, st, scene9, n, rows, cols, id_columns,
                       w.vis_words, w.skey, w.tiles, counts + 2);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, w.tiles, n_tiles, counts + 0);
    hipLaunchKernelGGL(k_scatter_keep, dim3(n_tiles), dim3(kThreads), 0, st, scene9, n, w.skey, w.tiles,
                       scene_out9);
    size_t sb = w.sort_bytes;
    R3D_HIP(rocprim::radix_sort_keys(w.sort_tmp, sb, w.skey, w.skey2, (size_t)n, 0, key_end, st));
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks_for(n * 9, kThreads, 2048)), dim3(kThreads), 0, st,
                       scene9, w.skey2, counts + 2, n, covered9);
  }
  if (m > 0) {
    int m_tiles = (int)((m + kTile - 1) / kTile);
    hipLaunchKernelGGL(k_mark, dim3(m_tiles), dim3(kThreads), 0, st, sample9, m, rows, cols, id_columns,
                       w.vis_words, w.mkey, (int32_t *)nullptr, counts + 1);
    size_t sb = w.sort_bytes;
    R3D_HIP(rocprim::radix_sort_keys(w.sort_tmp, sb, w.mkey, w.mkey2, (size_t)m, 0, key_end, st));
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks_for(m * 9, kThreads, 2048)), dim3(kThreads), 0, st,
                       sample9, w.mkey2, counts + 1, m, visible9);
  }
  R3D_LAUNCHED("occlusion_merge kernels");
  return R3D_OK;
}

int r3d_remove_space_for_spherical(const double *pcl9, int64_t n, float *xyzi, uint32_t *label,
                                   float *check, int32_t check_cols, void *stream) {
  if (n < 0 || (n > 0 && !pcl9) || (check && check_cols != 4 && check_cols != 5))
    return fail(R3D_E_ARG, "remove_space: bad argument");
  if (n == 0) return R3D_OK;
  hipLaunchKernelGGL(k_remove_space, dim3(blocks_for(n, kThreads, 2048)), dim3(kThreads), 0,
                     (hipStream_t)stream, pcl9, n, xyzi, label, check, check_cols);
  R3D_LAUNCHED("k_remove_space");
  return R3D_OK;
}

}  // extern "C"
