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

namespace r3d {

constexpr int kPT = 256;             // threads of the streaming kernels
constexpr int kPerThread = 8;
constexpr int kTile = kPT * kPerThread;   // points per block tile
constexpr int kST = 512;             // threads of the per-scene insert kernel
constexpr int kKeyCap = R3D_MAX_SAMPLE;
constexpr int kIdxBits = 13;         // kKeyCap == 1 << kIdxBits
constexpr int kRebaseRows = 16;      // block rows of the (normally idle) rebase launches

struct BatchWs {
  unsigned long long *qkeys;    // [B][2] ordered keys of min / max of z/r
  int32_t *tile_alive;          // [B*tiles]
  int32_t *tile_head;           // [B*tiles]
  int32_t *new_head;            // [B]
  int32_t *tail_tmp;            // [B*log_cap]
  uint32_t *cand;               // [B*npix] candidate pixel lists of the insert kernel
  int32_t *all_list;            // [B] identity
  int32_t *all_count;           // [1] = B
  int32_t *rebase_list;         // [B]
  int32_t *n_rebase;            // [1]
  int32_t *rebase_ticket;       // [1]
  unsigned long long *chunk_box; // [B*chunks] rows/cols bounding box of 64 consecutive points
  int32_t *n_proj;              // [B] points covered by the chunk boxes
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
  w.tile_head = c.take<int32_t>((size_t)b.B * tiles);
  w.new_head = c.take<int32_t>((size_t)b.B);
  w.tail_tmp = c.take<int32_t>((size_t)b.B * b.log_cap);
  w.cand = c.take<uint32_t>((size_t)b.B * npix);
  w.all_list = c.take<int32_t>((size_t)b.B);
  w.all_count = c.take<int32_t>(1);
  w.rebase_list = c.take<int32_t>((size_t)b.B);
  w.n_rebase = c.take<int32_t>(1);
  w.rebase_ticket = c.take<int32_t>(1);
  w.chunk_box = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.n_proj = c.take<int32_t>((size_t)b.B);
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

__device__ __forceinline__ bool point_alive(const r3d_batch_t &b, int s, int i, int n_head, int npix,
                                            int words) {
  int p = b.pix[(int64_t)s * b.cap + i];
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
}

__global__ void k_bounds_init(const int32_t *list, const int32_t *count, BatchWs w) {
  int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= *count) return;
  int s = list[li];
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

__global__ void k_bounds_finish(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w) {
  int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= *count) return;
  int s = list[li];
  unsigned long long kmin = w.qkeys[2 * s + 0], kmax = w.qkeys[2 * s + 1];
  if (kmin == ~0ull) {                    // no valid point: the reference raises (insertion.py:78)
    b.bounds[2 * s + 0] = b.bounds[2 * s + 1] = 0.0;
    b.extreme_pix[2 * s + 0] = b.extreme_pix[2 * s + 1] = -1;
    atomicOr(&b.status[s], R3D_S_NONFINITE);
    return;
  }
  double max_el = acos(ordered_key_inv(kmin));   // insertion.py:79
  double min_el = acos(ordered_key_inv(kmax));   // insertion.py:78
  b.bounds[2 * s + 0] = max_el;
  b.bounds[2 * s + 1] = min_el;
  b.extreme_pix[2 * s + 0] = b.extreme_pix[2 * s + 1] = -1;   // recorded by the projection pass
}

// ---- step 0 / rebase: reset the per-scene visibility stamps -------------------------------------
__global__ void __launch_bounds__(kPT)
k_reset(r3d_batch_t b, const int32_t *list, const int32_t *count) {
  int cnt = *count;
  int64_t npix = (int64_t)b.rows * b.cols;
  int words = (int)((npix + 31) / 32);
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    uint32_t *st = reinterpret_cast<uint32_t *>(b.stamp + (int64_t)s * npix);   // npix is even
    uint32_t *ev = b.ever + (int64_t)s * words;
    for (int64_t p = blockIdx.x * (int64_t)kPT + threadIdx.x; p < npix / 2; p += (int64_t)gridDim.x * kPT) {
      st[p] = 0u;
      if (p < words) ev[p] = 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) b.n_far[s] = 0;
  }
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
struct BoxAcc {
  int rmin = 0xFFFF, rmax = 0, cmin = 0xFFFF, cmax = 0;
  __device__ __forceinline__ void add(int row, int col) {
    rmin = row < rmin ? row : rmin;
    rmax = row > rmax ? row : rmax;
    cmin = col < cmin ? col : cmin;
    cmax = col > cmax ? col : cmax;
  }
  __device__ __forceinline__ unsigned long long wave_pack() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      int t;
      t = __shfl_xor(rmin, o, 64); rmin = t < rmin ? t : rmin;
      t = __shfl_xor(rmax, o, 64); rmax = t > rmax ? t : rmax;
      t = __shfl_xor(cmin, o, 64); cmin = t < cmin ? t : cmin;
      t = __shfl_xor(cmax, o, 64); cmax = t > cmax ? t : cmax;
    }
    return pack_box(rmin, rmax, cmin, cmax);   // rmin > rmax: empty box
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
    if (sp.el == bn.max_el) b.extreme_pix[2 * s + 0] = p;   // any holder will do
    if (sp.el == bn.min_el) b.extreme_pix[2 * s + 1] = p;
    if (sp.r > R3D_EMPTY_DEPTH) {           // "first hit overwrites the 500": insertion.py:122-125
      int f = atomicAdd(&b.n_far[s], 1);
      if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = p;
      else flags |= R3D_S_FAR_OVERFLOW;
    }
  }
  return p;
}

__global__ void __launch_bounds__(kPT)
k_project(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int chunks) {
  int cnt = *count;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    int t0 = blockIdx.x * kTile;
    if (blockIdx.x == 0 && threadIdx.x == 0) w.n_proj[s] = n;
    if (t0 >= n) continue;
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    int flags = 0;
#pragma unroll 2
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      BoxAcc box;
      if (i < n) {
        double x, y, z;
        load_point(b, s, i, n_head, x, y, z);
        b.pix[(int64_t)s * b.cap + i] = project_point(b, s, bn, x, y, z, flags, box);
      }
      unsigned long long packed = box.wave_pack();
      int i0 = t0 + k * kPT + (threadIdx.x & ~63);
      if ((threadIdx.x & 63) == 0 && i0 < n) w.chunk_box[(int64_t)s * chunks + (i0 >> 6)] = packed;
    }
    flags = wave_or_i32(flags);
    if ((threadIdx.x & 63) == 0 && flags) atomicOr(&b.status[s], flags);
  }
}

// ---- one insert candidate per scene ------------------------------------------------------------
// Row-aligned bit images (cols % 32 == 0, wpr = cols / 32 words per row) live in LDS; dilation and
// erosion with the 5-row x 3-column element of closing.py:20 are word-parallel shifts and ORs/ANDs,
// windows clipped at the image border exactly like the reference's (no azimuth wrap).
struct BitImage {
  uint32_t *w;
  int wpr, rows;
  __device__ __forceinline__ bool get(int p) const { return (w[p >> 5] >> (p & 31)) & 1u; }
  __device__ __forceinline__ void set(int p) { atomicOr(&w[p >> 5], 1u << (p & 31)); }
  __device__ __forceinline__ uint32_t word(int r, int j) const { return w[r * wpr + j]; }
};

// OR of a word with its two horizontal neighbours' bits (columns c-1, c, c+1), clipped at the row ends.
__device__ __forceinline__ uint32_t hor3(const BitImage &m, int r, int j) {
  uint32_t c = m.word(r, j);
  uint32_t l = j > 0 ? m.word(r, j - 1) : 0u;
  uint32_t rr = j < m.wpr - 1 ? m.word(r, j + 1) : 0u;
  return c | (c << 1) | (l >> 31) | (c >> 1) | (rr << 31);
}
// AND of the same three columns; a neighbour outside the image does not constrain (erosion border).
__device__ __forceinline__ uint32_t hand3(const BitImage &m, int r, int j) {
  uint32_t c = m.word(r, j);
  uint32_t l = j > 0 ? (m.word(r, j - 1) >> 31) : 1u;
  uint32_t rr = j < m.wpr - 1 ? (m.word(r, j + 1) << 31) : 0x80000000u;
  return c & ((c << 1) | l) & ((c >> 1) | rr);
}

// closing.py:44-57: all (up to 15) neighbour loads are issued first and are independent; an empty
// pixel holds R3D_SENT, so occupancy is read off the value.  The sum then runs drow outer,
// dcolumn inner over the occupied ones, as the reference's does.
template <class Load>
__device__ __forceinline__ double mean_of_occupied(const Load &load, int r, int c, int rows, int cols) {
  unsigned long long v[15];
#pragma unroll
  for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
    for (int dc = -1; dc <= 1; ++dc) {
      int rr = r + dr, cc = c + dc;
      bool in = rr >= 0 && rr < rows && cc >= 0 && cc < cols;
      v[(dr + 2) * 3 + (dc + 1)] = in ? load(rr * cols + cc) : R3D_SENT;
    }
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

__global__ void __launch_bounds__(kST)
k_insert(r3d_batch_t b, const double *__restrict__ samples5, const int64_t *__restrict__ sample_off,
         const int32_t *__restrict__ min_points, const int32_t *__restrict__ active, int step,
         int32_t *__restrict__ n_visible, int32_t *__restrict__ accepted, BatchWs w, int chunks) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int s = blockIdx.x;
  const int tid = threadIdx.x;
  const int rows = b.rows, cols = b.cols;
  const int npix = rows * cols;
  const int words = npix >> 5, wpr = cols >> 5;
  uint32_t *s_keys = reinterpret_cast<uint32_t *>(smem);          // [kKeyCap] sorted (pixel, index)
  uint32_t *s_img = s_keys + kKeyCap;                             // 5 bit images of `words` words
  uint32_t *s_oob = s_img + 5 * words;                            // [kKeyCap/32] el outside bounds
  int *s_misc = reinterpret_cast<int *>(s_oob + kKeyCap / 32);
  int *s_nvalid = s_misc + 0, *s_ncand = s_misc + 1, *s_rebase = s_misc + 2, *s_flags = s_misc + 3;
  int *s_rmin = s_misc + 4, *s_rmax = s_misc + 5;                 // sample row range
  int *s_cmin = s_misc + 6, *s_cmax = s_misc + 8;                 // [2] column range per image half
  int *s_nlist = s_misc + 10;                                     // chunks that touch the window
  int *s_scan = s_misc + 11;                                      // [kST/64 + 1]
  BitImage A{s_img, wpr, rows};                // sample occupancy
  BitImage T{s_img + words, wpr, rows};        // scratch: dilations, candidate mask, then visible pixels
  BitImage Cs{s_img + 2 * words, wpr, rows};   // sample closed
  BitImage D{s_img + 3 * words, wpr, rows};    // scene occupancy (window only)
  BitImage E{s_img + 4 * words, wpr, rows};    // scene closed (window only)

  const int64_t off = sample_off[s];
  const int64_t m64 = sample_off[s + 1] - off;
  const bool on = (!active || active[s]) && m64 > 0 && m64 <= kKeyCap;
  if (!on) {
    if (tid == 0) {
      n_visible[s] = 0;
      accepted[s] = 0;
      if (m64 > kKeyCap && (!active || active[s])) atomicOr(&b.status[s], R3D_S_SAMPLE_TOO_LARGE);
    }
    return;
  }
  const int m = (int)m64;
  int pw = 64;
  while (pw < m) pw <<= 1;

  for (int i = tid; i < 5 * words + kKeyCap / 32; i += kST) s_img[i] = 0u;
  if (tid < 11) s_misc[tid] = (tid == 4 || tid == 6 || tid == 7) ? 0x7FFFFFFF : (tid == 5 || tid == 8 || tid == 9) ? -1 : 0;
  __syncthreads();

  const Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], rows, cols);
  unsigned long long *grid = (unsigned long long *)b.grid + (int64_t)s * npix;
  unsigned long long *sgrid = (unsigned long long *)b.sgrid + (int64_t)s * npix;
  const double *rows5 = samples5 + off * 5;
  auto ld_scene = [&](int q) {
    return __hip_atomic_load(&grid[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ld_sample = [&](int q) {
    return __hip_atomic_load(&sgrid[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

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
            int p = row * cols + col;
            key = ((uint32_t)p << kIdxBits) | (uint32_t)j;
            atomicMin(&sgrid[p], depth_key(sp.r));
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

  // -- 2. sort (pixel, sample index): the order of visible_sample (insertion.py:474-482) ---------
  for (int k = 2; k <= pw; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < pw; i += kST) {
        int ixj = i ^ j;
        if (ixj > i) {
          uint32_t a = s_keys[i], c = s_keys[ixj];
          bool up = (i & k) == 0;
          if ((a > c) == up) {
            s_keys[i] = c;
            s_keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  const int nvalid = *s_nvalid;
  const int n_far = b.n_far[s] < R3D_FAR_CAP ? b.n_far[s] : R3D_FAR_CAP;

  // -- 3. bit images: sample occupancy, scene occupancy in the window around the sample ----------
  // candidates lie within 2 rows / 1 column of a sample pixel; their closing looks 4 rows / 2
  // columns further.  The window is rows [r_lo, r_hi] x one or two column intervals (whole words
  // jl[k]..jh[k]).  Pixels deeper than 500 m (far list) can be visible anywhere: whole image.
  int r_lo = 0, r_hi = rows - 1, n_iv = 1;
  int jl[2] = {0, 0}, jh[2] = {wpr - 1, 0};
  if (n_far == 0 && nvalid == 0) {
    r_hi = -1;                                           // nothing can be visible: empty window
  } else if (n_far == 0) {
    r_lo = *s_rmin - 6 < 0 ? 0 : *s_rmin - 6;
    r_hi = *s_rmax + 6 > rows - 1 ? rows - 1 : *s_rmax + 6;
    n_iv = 0;
    for (int h = 0; h < 2; ++h) {
      if (s_cmax[h] < 0) continue;
      int lo = s_cmin[h] - 3 < 0 ? 0 : s_cmin[h] - 3;
      int hi = s_cmax[h] + 3 > cols - 1 ? cols - 1 : s_cmax[h] + 3;
      int a = lo >> 5, z = hi >> 5;
      if (n_iv == 1 && a <= jh[0] + 1) jh[0] = z > jh[0] ? z : jh[0];   // touches the first: merge
      else {
        jl[n_iv] = a;
        jh[n_iv] = z;
        ++n_iv;
      }
    }
  }
  const int nj0 = jh[0] - jl[0] + 1, nj1 = n_iv > 1 ? jh[1] - jl[1] + 1 : 0, njw = nj0 + nj1;
  const int nrw = r_hi - r_lo + 1;
  // window word index e in [0, nrw * njw) -> (row, word-in-row)
  auto win_row = [&](int e) { return r_lo + e / njw; };
  auto win_word = [&](int e) {
    int k = e % njw;
    return k < nj0 ? jl[0] + k : jl[1] + (k - nj0);
  };
  for (int k = tid; k < nvalid; k += kST) {
    int p = (int)(s_keys[k] >> kIdxBits);
    if (k == 0 || (int)(s_keys[k - 1] >> kIdxBits) != p) A.set(p);
  }
  // -- 3b. the scene's range image inside the window, built from the points (insertion.py:118-125)
  // Alive points whose pixel lies in the window min-reduce into the (all-empty) scratch image:
  // first the 64-point chunks whose bounding box touches the window, then the points appended
  // since the last projection.  Dead points (their pixel was visible at a later step) are skipped,
  // which is what culling them (:472-473) does to the image.
  uint32_t *cand = w.cand + (int64_t)s * npix;
  const int n_total = b.n_total[s], n_head = b.n_head[s], n_log = b.n_log[s];
  auto in_window = [&](int p) {
    int r = p / cols, j = (p - r * cols) >> 5;
    return r >= r_lo && r <= r_hi && ((j >= jl[0] && j <= jh[0]) || (n_iv > 1 && j >= jl[1] && j <= jh[1]));
  };
  auto reduce_point = [&](int i) {
    int p = b.pix[(int64_t)s * b.cap + i];
    if (!in_window(p) || !point_alive(b, s, i, n_head, npix, words)) return;
    double x, y, z;
    load_point(b, s, i, n_head, x, y, z);
    atomicMin(&grid[p], depth_key(sqrt(x * x + y * y + z * z)));
  };
  if (nrw > 0) {
    const int n_proj = w.n_proj[s] < n_total ? w.n_proj[s] : n_total;
    const int n_chunks = (n_proj + 63) >> 6;
    const unsigned long long *boxes = w.chunk_box + (int64_t)s * chunks;
    for (int c = tid; c < n_chunks; c += kST) {
      unsigned long long bx = boxes[c];
      int rmin = (int)(bx & 0xFFFF), rmax = (int)((bx >> 16) & 0xFFFF);
      int jmin = (int)((bx >> 32) & 0xFFFF) >> 5, jmax = (int)((bx >> 48) & 0xFFFF) >> 5;
      bool hit = rmin <= r_hi && rmax >= r_lo &&
                 ((jmin <= jh[0] && jmax >= jl[0]) || (n_iv > 1 && jmin <= jh[1] && jmax >= jl[1]));
      if (hit) cand[atomicAdd(s_nlist, 1)] = (uint32_t)c;
    }
    __syncthreads();
    const int nlist = *s_nlist;
    for (int e = tid; e < nlist * 64; e += kST) {
      int i = (int)(cand[e >> 6] << 6) + (e & 63);
      if (i < n_proj) reduce_point(i);
    }
    for (int i = n_proj + tid; i < n_total; i += kST) reduce_point(i);
  }
  __syncthreads();     // every atomic of the block has been performed (vmcnt(0) at the barrier)

  if (nrw > 0) {
    // scene occupancy: each lane loads the 4 pixels of one nibble of a window word, 8 lanes
    // assemble a word by shuffles -> plain LDS store, no LDS atomics
    const int total = nrw * njw * 8;
    for (int e0 = 0; e0 < total; e0 += kST) {
      int e = e0 + tid;
      uint32_t nib = 0;
      int r = 0, j = 0;
      if (e < total) {
        r = win_row(e >> 3);
        j = win_word(e >> 3);
        int q0 = r * cols + (j << 5) + ((e & 7) << 2);
        nib = (ld_scene(q0) != R3D_SENT ? 1u : 0u) | (ld_scene(q0 + 1) != R3D_SENT ? 2u : 0u) |
              (ld_scene(q0 + 2) != R3D_SENT ? 4u : 0u) | (ld_scene(q0 + 3) != R3D_SENT ? 8u : 0u);
        nib <<= (e & 7) << 2;
      }
      nib |= __shfl_xor(nib, 1, 64);
      nib |= __shfl_xor(nib, 2, 64);
      nib |= __shfl_xor(nib, 4, 64);
      if (e < total && (e & 7) == 0) D.w[r * wpr + j] = nib;
    }
  }
  __syncthreads();

  // -- 4. closing of both occupancies (closing.py:9-23) by word-parallel dilate / erode ----------
  // Exact on every row at least 2 inside the window (or at the image border): candidates are.
  for (int pass = 0; pass < 4; ++pass) {
    const BitImage &src = pass == 0 ? A : pass == 2 ? D : T;
    BitImage &dst = pass == 0 ? T : pass == 1 ? Cs : pass == 2 ? T : E;
    const bool erode = pass & 1;
    for (int e = tid; e < nrw * njw; e += kST) {
      int r = win_row(e), j = win_word(e);
      uint32_t acc = erode ? 0xFFFFFFFFu : 0u;
      for (int dr = -2; dr <= 2; ++dr) {
        int rr = r + dr;
        if (rr < 0 || rr >= rows) continue;
        if (erode) acc &= hand3(src, rr, j);
        else acc |= hor3(src, rr, j);
      }
      dst.w[r * wpr + j] = acc;
    }
    __syncthreads();
  }

  // -- 5. candidate pixels: where the sample is closed, plus the far neighbourhoods --------------
  for (int e = tid; e < nrw * njw; e += kST) {
    int idx = win_row(e) * wpr + win_word(e);
    T.w[idx] = Cs.w[idx];
  }
  __syncthreads();
  for (int f = tid; f < n_far; f += kST) {
    int p = b.far_pix[(int64_t)s * R3D_FAR_CAP + f];
    int r = p / cols, c = p - r * cols;
    for (int dr = -2; dr <= 2; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        if (rr >= 0 && rr < rows && cc >= 0 && cc < cols) T.set(rr * cols + cc);
      }
  }
  __syncthreads();
  for (int e = tid; e < nrw * njw; e += kST) {
    int r = win_row(e), j = win_word(e);
    uint32_t bits = T.w[r * wpr + j];
    if (!bits) continue;
    int pos = atomicAdd(s_ncand, __popc(bits));
    while (bits) {
      int bit = __ffs(bits) - 1;
      bits &= bits - 1;
      cand[pos++] = (uint32_t)(r * cols + (j << 5) + bit);
    }
  }
  __syncthreads();
  const int ncand = *s_ncand;
  for (int e = tid; e < nrw * njw; e += kST) T.w[win_row(e) * wpr + win_word(e)] = 0u;
  __syncthreads();
  BitImage &vis = T;

  // -- 6. visibility on the candidates: smoothed sample depth < smoothed scene depth (:461-467) --
  for (int ci = tid; ci < ncand; ci += kST) {
    int q = (int)cand[ci];
    int r = q / cols, c = q - r * cols;
    double sd = R3D_EMPTY_DEPTH, cd = R3D_EMPTY_DEPTH;
    if (A.get(q)) sd = key_depth(ld_sample(q));
    else if (Cs.get(q)) sd = mean_of_occupied(ld_sample, r, c, rows, cols);
    if (D.get(q)) cd = key_depth(ld_scene(q));
    else if (E.get(q)) cd = mean_of_occupied(ld_scene, r, c, rows, cols);
    if (sd < cd) vis.set(q);
  }
  __syncthreads();

  // -- 7. count the visible sample points, accept test (insertion.py:511-517) --------------------
  int mine = 0;
  for (int k = tid; k < nvalid; k += kST) mine += vis.get((int)(s_keys[k] >> kIdxBits)) ? 1 : 0;
  int nvis;
  (void)block_escan_i32(mine, s_scan, nvis);
  int need = min_points[s];
  bool accept = nvis > 0 && nvis >= need;
  if (accept && ((int64_t)n_total + nvis > b.cap || (int64_t)n_log + nvis > b.log_cap)) {
    accept = false;
    if (tid == 0) atomicOr(&b.status[s], R3D_S_CAPACITY);
  }

  // -- 8. commit: append (insertion.py:526), patch the range image, stamp ------------------------
  if (accept) {
    int base = 0;
    for (int k0 = 0; k0 < nvalid; k0 += kST) {
      int k = k0 + tid;
      uint32_t key = k < nvalid ? s_keys[k] : 0u;
      int p = (int)(key >> kIdxBits);
      int flag = (k < nvalid && vis.get(p)) ? 1 : 0;
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
        b.pix[(int64_t)s * b.cap + dst] = p;
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
    const int pix_of_max = b.extreme_pix[2 * s + 0], pix_of_min = b.extreme_pix[2 * s + 1];
    uint16_t *stamp = b.stamp + (int64_t)s * npix;
    uint32_t *ever = b.ever + (int64_t)s * words;
    for (int ci = tid; ci < ncand; ci += kST) {
      int q = (int)cand[ci];
      if (!vis.get(q)) continue;
      if (q == pix_of_max || q == pix_of_min) *s_rebase = 1;   // the recorded extreme point is culled
      unsigned long long nv = A.get(q) ? ld_sample(q) : R3D_SENT;
      if (nv != R3D_SENT && key_depth(nv) > R3D_EMPTY_DEPTH) {
        int f = atomicAdd(&b.n_far[s], 1);
        if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = q;
        else atomicOr(s_flags, R3D_S_FAR_OVERFLOW);
      }
      stamp[q] = (uint16_t)step;
      atomicOr(&ever[q >> 5], 1u << (q & 31));
    }
  }
  __syncthreads();

  // -- 9. leave both scratch images clean (all-empty), publish ----------------------------------
  for (int e = tid; e < nrw * njw * 8; e += kST) {
    int q0 = win_row(e >> 3) * cols + (win_word(e >> 3) << 5) + ((e & 7) << 2);
    ulonglong2 sent = make_ulonglong2(R3D_SENT, R3D_SENT);
    reinterpret_cast<ulonglong2 *>(grid + q0)[0] = sent;
    reinterpret_cast<ulonglong2 *>(grid + q0)[1] = sent;
  }
  for (int k = tid; k < nvalid; k += kST) {
    int p = (int)(s_keys[k] >> kIdxBits);
    if (k == 0 || (int)(s_keys[k - 1] >> kIdxBits) != p) sgrid[p] = R3D_SENT;
  }
  if (tid == 0) {
    n_visible[s] = nvis;
    accepted[s] = accept ? 1 : 0;
    if (*s_flags) atomicOr(&b.status[s], *s_flags);
    if (accept) {
      b.n_total[s] = n_total + nvis;
      b.n_log[s] = n_log + nvis;
      if (*s_rebase) {
        b.rebase[s] += 1;                                       // single writer per scene
        w.rebase_list[atomicAdd(w.n_rebase, 1)] = s;
      }
    }
  }
}

// ---- compaction: drop dead points (finish, or rebase) ------------------------------------------
__global__ void __launch_bounds__(kPT)
k_alive_count(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles) {
  __shared__ int s_a[kPT / 64], s_h[kPT / 64];
  int cnt = *count;
  int npix = b.rows * b.cols, words = (npix + 31) / 32;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    int t0 = blockIdx.x * kTile;
    int alive = 0, head = 0;
    if (t0 < n) {
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) {
        int i = t0 + k * kPT + threadIdx.x;
        if (i < n && point_alive(b, s, i, n_head, npix, words)) {
          ++alive;
          head += i < n_head ? 1 : 0;
        }
      }
    }
    alive = wave_sum_i32(alive);
    head = wave_sum_i32(head);
    if ((threadIdx.x & 63) == 0) {
      s_a[threadIdx.x >> 6] = alive;
      s_h[threadIdx.x >> 6] = head;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int a = 0, h = 0;
      for (int v = 0; v < kPT / 64; ++v) {
        a += s_a[v];
        h += s_h[v];
      }
      w.tile_alive[(int64_t)s * tiles + blockIdx.x] = a;
      w.tile_head[(int64_t)s * tiles + blockIdx.x] = h;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(1024)
k_alive_scan(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles) {
  __shared__ int sm[1024 / 64 + 1];
  __shared__ int s_carry, s_head;
  int cnt = *count;
  for (int li = blockIdx.x; li < cnt; li += gridDim.x) {
    int s = list[li];
    if (threadIdx.x == 0) s_carry = s_head = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
      int t = base + threadIdx.x;
      int a = t < tiles ? w.tile_alive[(int64_t)s * tiles + t] : 0;
      int h = t < tiles ? w.tile_head[(int64_t)s * tiles + t] : 0;
      int tot, htot;
      int ex = block_escan_i32(a, sm, tot);
      (void)block_escan_i32(h, sm, htot);
      int carry = s_carry;
      if (t < tiles) w.tile_alive[(int64_t)s * tiles + t] = carry + ex;
      __syncthreads();
      if (threadIdx.x == 0) {
        s_carry = carry + tot;
        s_head += htot;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      b.n_out[s] = s_carry;
      w.new_head[s] = s_head;
    }
    __syncthreads();
  }
}

// Survivors in original order (insertion.py:472-473 applied once for all steps), float4 + label
// straight into the output arrays; tail references follow into tail_tmp for a rebase.
__global__ void __launch_bounds__(kPT)
k_alive_write(r3d_batch_t b, const int32_t *list, const int32_t *count, BatchWs w, int tiles) {
  __shared__ int sm[kPT / 64 + 1];
  int cnt = *count;
  int npix = b.rows * b.cols, words = (npix + 31) / 32;
  for (int li = blockIdx.y; li < cnt; li += gridDim.y) {
    int s = list[li];
    int n = b.n_total[s], n_head = b.n_head[s];
    int t0 = blockIdx.x * kTile;
    if (t0 >= n) continue;
    int base = w.tile_alive[(int64_t)s * tiles + blockIdx.x];
    int new_head = w.new_head[s];
    const float4 *src = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    float4 *dst = reinterpret_cast<float4 *>(b.out_xyzi) + (int64_t)s * b.cap;
    for (int k = 0; k < kPerThread; ++k) {
      int i = t0 + k * kPT + threadIdx.x;
      int flag = (i < n && point_alive(b, s, i, n_head, npix, words)) ? 1 : 0;
      int tot;
      int ex = block_escan_i32(flag, sm, tot);
      if (flag) {
        int o = base + ex;
        dst[o] = src[i];
        b.out_label[(int64_t)s * b.cap + o] = b.label[(int64_t)s * b.cap + i];
        if (i >= n_head)
          w.tail_tmp[(int64_t)s * b.log_cap + (o - new_head)] =
              b.tail_ref[(int64_t)s * b.log_cap + (i - n_head)];
      }
      base += tot;
    }
  }
}

// ---- rebase: one workgroup re-bases one flagged scene (rare path) --------------------------------
// Triggered when an accepted insert may have moved the elevation bounds (k_insert, step 8).  Does,
// for that scene only, what the reference does for every insert (insertion.py:373-375): drop the
// culled points, recompute the bounds, re-project every point.  All phases run inside one block so
// the idle case costs one empty launch; phases are separated by a device-scope fence + barrier
// because later phases re-read what earlier ones wrote.
constexpr int kRB = 1024;

__device__ __forceinline__ void phase_sync() {
  __threadfence();
  __syncthreads();
}

__global__ void __launch_bounds__(kRB)
k_rebase(r3d_batch_t b, BatchWs w, int chunks) {
  __shared__ int sm[kRB / 64 + 1];
  __shared__ unsigned long long s_min[kRB / 64], s_max[kRB / 64];
  const int tid = threadIdx.x;
  const int npix = b.rows * b.cols, words = (npix + 31) / 32;
  const int cnt = *w.n_rebase;
  for (int li = blockIdx.x; li < cnt; li += gridDim.x) {
    const int s = w.rebase_list[li];
    const int n = b.n_total[s], n_head = b.n_head[s];
    float4 *xyzi = reinterpret_cast<float4 *>(b.xyzi) + (int64_t)s * b.cap;
    uint32_t *label = b.label + (int64_t)s * b.cap;
    int32_t *tref = b.tail_ref + (int64_t)s * b.log_cap;
    // (a) alive head points: the new n_head
    int mine = 0;
    for (int i = tid; i < n_head; i += kRB) mine += point_alive(b, s, i, n_head, npix, words) ? 1 : 0;
    int new_head;
    (void)block_escan_i32(mine, sm, new_head);
    // (b) in-place stable compaction, tile by tile (write index <= read index)
    int base = 0;
    for (int t0 = 0; t0 < n; t0 += kRB) {
      int i = t0 + tid;
      int flag = (i < n && point_alive(b, s, i, n_head, npix, words)) ? 1 : 0;
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t lab = 0;
      int tr = 0;
      if (flag) {
        p = xyzi[i];
        lab = label[i];
        if (i >= n_head) tr = tref[i - n_head];
      }
      int tot;
      int ex = block_escan_i32(flag, sm, tot);      // barriers: every read of the tile is done
      if (flag) {
        int o = base + ex;
        xyzi[o] = p;
        label[o] = lab;
        if (i >= n_head) tref[o - new_head] = tr;
      }
      base += tot;
      phase_sync();
    }
    const int n_new = base;
    if (tid == 0) {
      b.n_head[s] = new_head;
      b.n_total[s] = n_new;
    }
    phase_sync();
    // (c) bounds (insertion.py:78-79) via the extreme z/r
    unsigned long long lmin = ~0ull, lmax = 0ull;
    int bad = 0;
    for (int i = tid; i < n_new; i += kRB) {
      double x, y, z;
      load_point(b, s, i, new_head, x, y, z);
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
      b.extreme_pix[2 * s + 0] = b.extreme_pix[2 * s + 1] = -1;
      b.n_far[s] = 0;
    }
    // (d) reset the visibility stamps
    for (int p = tid; p < npix; p += kRB) {
      b.stamp[(int64_t)s * npix + p] = 0;
      if (p < words) b.ever[(int64_t)s * words + p] = 0u;
    }
    phase_sync();
    // (e) re-project (insertion.py:74-76, :104-116): pixel ids and chunk boxes
    Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], b.rows, b.cols);
    int flags = 0;
    for (int i0 = 0; i0 < n_new; i0 += kRB) {
      int i = i0 + tid;
      BoxAcc box;
      if (i < n_new) {
        double x, y, z;
        load_point(b, s, i, new_head, x, y, z);
        b.pix[(int64_t)s * b.cap + i] = project_point(b, s, bn, x, y, z, flags, box);
      }
      unsigned long long packed = box.wave_pack();
      int c0 = i0 + (tid & ~63);
      if ((tid & 63) == 0 && c0 < n_new) w.chunk_box[(int64_t)s * chunks + (c0 >> 6)] = packed;
    }
    if (tid == 0) w.n_proj[s] = n_new;
    if (flags) atomicOr(&b.status[s], flags);
    phase_sync();
  }
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
  if (b->cap > (int64_t)1 << 30 || (int64_t)b->rows * b->cols > (int64_t)1 << (32 - kIdxBits))
    return fail(R3D_E_ARG, "batch: cap or range image too large for 32-bit point / pixel ids");
  if (!b->xyzi || !b->label || !b->pix || !b->n_head || !b->n_total || !b->tail_ref || !b->log5 ||
      !b->log_birth || !b->n_log || !b->grid || !b->sgrid || !b->stamp || !b->ever || !b->bounds ||
      !b->extreme_pix || !b->far_pix || !b->n_far || !b->rebase || !b->status || !b->out_xyzi ||
      !b->out_label || !b->n_out || !b->workspace)
    return fail(R3D_E_ARG, "batch: null array");
  if (b->cols % 32 != 0)
    return fail(R3D_E_ARG, "batch: cols must be a multiple of 32 (row-aligned bit images)");
  size_t lds = (size_t)kKeyCap * 4 + 5 * (size_t)mask_words(*b) * 4 + kKeyCap / 8 + 64 * 4;
  if (lds > 160 * 1024)
    return fail(R3D_E_ARG, "batch: range image too large for the LDS-resident masks of k_insert");
  if (b->workspace_bytes < carve_batch(*b, nullptr).total)
    return fail(R3D_E_WORKSPACE, "batch: workspace smaller than r3d_batch_workspace_bytes()");
  return R3D_OK;
}

static size_t insert_lds_bytes(const r3d_batch_t &b) {
  return (size_t)kKeyCap * 4 + 5 * (size_t)mask_words(b) * 4 + kKeyCap / 8 + 64 * 4;
}

// bounds -> reset -> project for the scenes of (list, count); rows = block rows of the launches.
static int launch_reproject(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                            const int32_t *count, int rows, hipStream_t st) {
  int tiles = tiles_of(b);
  int lb = (b.B + 255) / 256;
  hipLaunchKernelGGL(k_bounds_init, dim3(lb), dim3(256), 0, st, list, count, w);
  hipLaunchKernelGGL(k_bounds, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w);
  hipLaunchKernelGGL(k_bounds_finish, dim3(lb), dim3(256), 0, st, b, list, count, w);
  int64_t npix = (int64_t)b.rows * b.cols;
  int rb = (int)((npix + kPT * 4 - 1) / (kPT * 4));
  hipLaunchKernelGGL(k_reset, dim3(rb, rows), dim3(kPT), 0, st, b, list, count);
  hipLaunchKernelGGL(k_project, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, chunks_of(b));
  R3D_LAUNCHED("reproject kernels");
  return R3D_OK;
}

static int launch_compact(const r3d_batch_t &b, const BatchWs &w, const int32_t *list,
                          const int32_t *count, int rows, hipStream_t st) {
  int tiles = tiles_of(b);
  hipLaunchKernelGGL(k_alive_count, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, tiles);
  hipLaunchKernelGGL(k_alive_scan, dim3(rows), dim3(1024), 0, st, b, list, count, w, tiles);
  hipLaunchKernelGGL(k_alive_write, dim3(tiles, rows), dim3(kPT), 0, st, b, list, count, w, tiles);
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

int r3d_batch_elev_bounds(const r3d_batch_t *b, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  int tiles = tiles_of(*b), lb = (b->B + 255) / 256;
  hipLaunchKernelGGL(k_bounds_init, dim3(lb), dim3(256), 0, st, w.all_list, w.all_count, w);
  hipLaunchKernelGGL(k_bounds, dim3(tiles, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w);
  hipLaunchKernelGGL(k_bounds_finish, dim3(lb), dim3(256), 0, st, *b, w.all_list, w.all_count, w);
  R3D_LAUNCHED("bounds kernels");
  return R3D_OK;
}

int r3d_batch_project(const r3d_batch_t *b, void *stream) {
  int rc = check_batch(b);
  if (rc != R3D_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  BatchWs w = carve_batch(*b, b->workspace);
  int64_t npix = (int64_t)b->rows * b->cols;
  int rb = (int)((npix + kPT * 4 - 1) / (kPT * 4));
  hipLaunchKernelGGL(k_reset, dim3(rb, b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count);
  hipLaunchKernelGGL(k_project, dim3(tiles_of(*b), b->B), dim3(kPT), 0, st, *b, w.all_list, w.all_count, w,
                     chunks_of(*b));
  R3D_LAUNCHED("project kernels");
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
