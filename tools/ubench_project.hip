// Micro-benchmark (not part of the library): where does the projection pass spend its time?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ubench tools/ubench_project.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../pcl-augmentation_amd/csrc/r3d_device.hpp"
using namespace r3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int B = 256, N = 120000, ROWS = 112, COLS = 1440, NPIX = ROWS * COLS;

__global__ void gen(float4 *xyzi) {   // ring-major synthetic scan, 64 beams x 1875 az
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * N) return;
  int k = i % N, ring = k / 1875, a = k % 1875;
  unsigned h = (unsigned)i * 2654435761u;
  double el = (-24.8 + 26.8 * ring / 63.0) * 0.017453292519943295 + ((h & 1023) - 512) * 4e-7;
  double az = -kPi + kTwoPi * a / 1875.0 + (((h >> 10) & 1023) - 512) * 2e-7;
  double rad = el < -0.02 ? fmin(1.73 / sin(-el), 60.0) : 40.0;
  xyzi[i] = make_float4((float)(rad * cos(el) * cos(az)), (float)(rad * cos(el) * sin(az)), (float)(rad * sin(el)), 0.5f);
}

template <int V>
__global__ void __launch_bounds__(256) proj(const float4 *__restrict__ xyzi, int *__restrict__ pix,
                                            unsigned long long *__restrict__ grid, double max_el, double min_el) {
  int s = blockIdx.y;
  int t0 = blockIdx.x * 2048;
  Binning bn = make_binning(max_el, min_el, ROWS, COLS);
  unsigned long long *g = grid + (int64_t)s * NPIX;
  const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az);
  const float elo = (float)(bn.min_el + 0.00001);
#pragma unroll 2
  for (int k = 0; k < 8; ++k) {
    int i = t0 + k * 256 + threadIdx.x;
    if (i >= N) continue;
    float4 p = xyzi[(int64_t)s * N + i];
    int px = 0;
    unsigned long long key = 0;
    if (V == 9) {
    } else if (V == 0 || V == 1) {              // full float64 math
      Sph sp = spherical((double)p.x, (double)p.y, (double)p.z);
      int row, col;
      bin_point(bn, sp.az, sp.el, row, col);
      px = row * COLS + col;
      key = depth_key(sp.r);
    } else if (V == 2) {                 // no math
      px = (int)(((unsigned)i * 1u) % NPIX);
      key = (unsigned long long)i;
    } else if (V == 10) {                // no math, sequential with duplicates (3 pixels per 4 points)
      px = (int)(((unsigned)i * 3u / 4u) % NPIX);
      key = (unsigned long long)i;
    } else if (V == 11) {                // no math, no loads: the ring -> row / column map of a real scan
      int ring = i / 1875, a = i - ring * 1875;
      px = (ring * 111 / 63) * COLS + a * COLS / 1875;
      key = (unsigned long long)i;
    } else if (V == 12) {                // same map, columns only (all rings in consecutive rows)
      int ring = i / 1875, a = i - ring * 1875;
      px = ring * COLS + a * COLS / 1875;
      key = (unsigned long long)i;
    } else {                             // float32 fast path, float64 only near a bin edge; s-key (V3/4/7); V5/6: never slow
      double x = p.x, y = p.y, z = p.z;
      double ss = x * x + y * y + z * z;
      float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
      float q = p.z * __frsqrt_rn(r2);
      float el = acosf(q), az = atan2f(p.y, p.x) + 3.14159265358979f;
      float tr = (el - elo) * inv_del, tc = az * inv_daz;
      float fr = tr - floorf(tr), fc = tc - floorf(tc);
      int row = (int)tr, col = (int)tc;
      bool slow = (V == 3 || V == 4 || V == 7 || V == 14) && (fr < 2e-3f || fr > 0.998f || fc < 2e-3f || fc > 0.998f || fabsf(q) > 0.9f || tr < 0.f);
      if (slow) {
        Sph sp = spherical(x, y, z);
        bin_point(bn, sp.az, sp.el, row, col);
      }
      row = row < 0 ? 0 : row > ROWS - 1 ? ROWS - 1 : row;
      col = col < 0 ? 0 : col > COLS - 1 ? COLS - 1 : col;
      px = row * COLS + col;
      key = depth_key(ss);
    }
    if (V == 0 || V == 2 || V == 3 || V == 5 || V == 10 || V == 11 || V == 12) atomicMin(&g[px], key);
    if (V == 8) {                        // fast path; unique pixels compacted to the low lanes (ascending, dense)
      __shared__ int s_px[4][64];
      __shared__ unsigned long long s_key[4][64];
      int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      for (int o = 1; o <= 4; o <<= 1) {
        unsigned long long k2 = __shfl_down(key, o, 64);
        int p2 = __shfl_down(px, o, 64);
        if (lane + o < 64 && p2 == px) key = k2 < key ? k2 : key;
      }
      int pp = __shfl_up(px, 1, 64);
      bool head = lane == 0 || pp != px || (lane & 7) == 0;
      unsigned long long m = __ballot(head);
      int pos = __popcll(m & ((1ull << lane) - 1));
      if (head) { s_px[wv][pos] = px; s_key[wv][pos] = key; }
      int nh = __popcll(m);
      if (lane < nh) atomicMin(&g[s_px[wv][lane]], s_key[wv][lane]);
    }
    if (V == 13 || V == 14) {            // exact run de-duplication: one atomic per run of equal pixels
      int lane = threadIdx.x & 63;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        unsigned long long k2 = __shfl_down(key, o, 64);
        int p2 = __shfl_down(px, o, 64);
        if (lane + o < 64 && p2 == px) key = k2 < key ? k2 : key;
      }
      int pp = __shfl_up(px, 1, 64);
      if (lane == 0 || pp != px) atomicMin(&g[px], key);
    }
    if (V == 9) {                        // atomics only for pixels (no math): real pixel ids from a previous run
      atomicMin(&g[pix[(int64_t)s * N + i]], (unsigned long long)i);
    }
    if (V == 4) {                        // wave de-duplication of equal adjacent pixels
      int lane = threadIdx.x & 63;
      for (int o = 1; o <= 4; o <<= 1) {
        unsigned long long k2 = __shfl_down(key, o, 64);
        int p2 = __shfl_down(px, o, 64);
        if (lane + o < 64 && p2 == px) key = k2 < key ? k2 : key;
      }
      int pp = __shfl_up(px, 1, 64);
      if (lane == 0 || pp != px || (lane & 7) == 0) atomicMin(&g[px], key);
    }
    if (V != 9 && (V < 10 || V > 12)) pix[(int64_t)s * N + i] = px;
  }
}

// loads of the whole tile first, then math + atomics: no load ever waits behind an atomic (vmcnt is in order)
template <int V>
__global__ void __launch_bounds__(256) proj_hoist(const float4 *__restrict__ xyzi, int *__restrict__ pix,
                                                  unsigned long long *__restrict__ grid, double max_el, double min_el) {
  int s = blockIdx.y;
  int t0 = blockIdx.x * 2048;
  Binning bn = make_binning(max_el, min_el, ROWS, COLS);
  unsigned long long *g = grid + (int64_t)s * NPIX;
  const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az);
  const float elo = (float)(bn.min_el + 0.00001);
  float4 pts[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int i = t0 + k * 256 + threadIdx.x;
    pts[k] = i < N ? xyzi[(int64_t)s * N + i] : make_float4(1.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int i = t0 + k * 256 + threadIdx.x;
    float4 p = pts[k];
    int px; unsigned long long key;
    if (V == 0) {
      Sph sp = spherical((double)p.x, (double)p.y, (double)p.z);
      int row, col;
      bin_point(bn, sp.az, sp.el, row, col);
      px = row * COLS + col;
      key = depth_key(sp.r);
    } else {
      double x = p.x, y = p.y, z = p.z;
      double ss = x * x + y * y + z * z;
      float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
      float q = p.z * __frsqrt_rn(r2);
      float el = acosf(q), az = atan2f(p.y, p.x) + 3.14159265358979f;
      float tr = (el - elo) * inv_del, tc = az * inv_daz;
      int row = (int)tr, col = (int)tc;
      row = row < 0 ? 0 : row > ROWS - 1 ? ROWS - 1 : row;
      col = col < 0 ? 0 : col > COLS - 1 ? COLS - 1 : col;
      px = row * COLS + col;
      key = depth_key(ss);
    }
    if (i < N) {
      atomicMin(&g[px], key);
      pix[(int64_t)s * N + i] = px;
    }
  }
}

// LDS-privatised range-image tile: the block's points min-reduce into 4 rows of LDS, then each
// wave flushes 64 consecutive pixels (8 whole 64-B lines) with one atomic instruction.
constexpr int WR = 4;
template <int V>
__global__ void __launch_bounds__(256) proj_tile(const float4 *__restrict__ xyzi, int *__restrict__ pix,
                                                 unsigned long long *__restrict__ grid, double max_el, double min_el) {
  __shared__ unsigned long long tile[WR * COLS];
  __shared__ int s_rmin;
  int s = blockIdx.y;
  int t0 = blockIdx.x * 2048;
  Binning bn = make_binning(max_el, min_el, ROWS, COLS);
  unsigned long long *g = grid + (int64_t)s * NPIX;
  const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az);
  const float elo = (float)(bn.min_el + 0.00001);
  if (threadIdx.x == 0) s_rmin = ROWS;
  for (int j = threadIdx.x; j < WR * COLS; j += 256) tile[j] = R3D_SENT;
  int px[8]; unsigned long long key[8];
  int rmin = ROWS;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int i = t0 + k * 256 + threadIdx.x;
    px[k] = -1;
    if (i >= N) continue;
    float4 p = xyzi[(int64_t)s * N + i];
    double x = p.x, y = p.y, z = p.z;
    int row, col;
    if (V == 0) {
      Sph sp = spherical(x, y, z);
      bin_point(bn, sp.az, sp.el, row, col);
      key[k] = depth_key(sp.r);
    } else {
      key[k] = depth_key(x * x + y * y + z * z);
      float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
      float q = p.z * __frsqrt_rn(r2);
      float el = acosf(q), az = atan2f(p.y, p.x) + 3.14159265358979f;
      row = (int)((el - elo) * inv_del); col = (int)(az * inv_daz);
      row = row < 0 ? 0 : row > ROWS - 1 ? ROWS - 1 : row;
      col = col < 0 ? 0 : col > COLS - 1 ? COLS - 1 : col;
    }
    px[k] = row * COLS + col;
    rmin = row < rmin ? row : rmin;
    pix[(int64_t)s * N + i] = px[k];
  }
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(rmin, o, 64); rmin = t < rmin ? t : rmin; }
  if ((threadIdx.x & 63) == 0) atomicMin(&s_rmin, rmin);
  __syncthreads();
  const int base = s_rmin * COLS;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (px[k] < 0) continue;
    int l = px[k] - base;
    if (l < WR * COLS) atomicMin(&tile[l], key[k]);
    else atomicMin(&g[px[k]], key[k]);
  }
  __syncthreads();
  int lim = NPIX - base < WR * COLS ? NPIX - base : WR * COLS;
  for (int j = threadIdx.x; j < lim; j += 256) {
    unsigned long long v = tile[j];
    if (v != R3D_SENT) atomicMin(&g[base + j], v);
  }
}

template <int V>
float run_tile(const float4 *x, int *pix, unsigned long long *grid, double mx, double mn, int reps = 10) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g((N + 2047) / 2048, B);
  proj_tile<V><<<g, 256>>>(x, pix, grid, mx, mn);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) proj_tile<V><<<g, 256>>>(x, pix, grid, mx, mn);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int V>
float run_hoist(const float4 *x, int *pix, unsigned long long *grid, double mx, double mn, int reps = 10) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g((N + 2047) / 2048, B);
  proj_hoist<V><<<g, 256>>>(x, pix, grid, mx, mn);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) proj_hoist<V><<<g, 256>>>(x, pix, grid, mx, mn);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int V>
float run(const float4 *x, int *pix, unsigned long long *grid, double mx, double mn, int reps = 10) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g((N + 2047) / 2048, B);
  hipMemset(grid, 0xFF, (size_t)B * NPIX * 8);
  proj<V><<<g, 256>>>(x, pix, grid, mx, mn);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) proj<V><<<g, 256>>>(x, pix, grid, mx, mn);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  float4 *x; int *pix, *pix2; unsigned long long *grid;
  CK(hipMalloc(&x, (size_t)B * N * 16)); CK(hipMalloc(&pix, (size_t)B * N * 4)); CK(hipMalloc(&pix2, (size_t)B * N * 4));
  CK(hipMalloc(&grid, (size_t)B * NPIX * 8));
  gen<<<(B * (int64_t)N + 255) / 256, 256>>>(x);
  CK(hipDeviceSynchronize());
  double mn = acos(sin(2.0 * 0.017453292519943295 + 3e-4)), mx = acos(sin(-24.8 * 0.017453292519943295 - 3e-4));
  printf("bounds %f %f\n", mx, mn);
  printf("V0 f64 + atomic        : %.3f ms\n", run<0>(x, pix, grid, mx, mn));
  printf("V1 f64, no atomic      : %.3f ms\n", run<1>(x, pix, grid, mx, mn));
  printf("V2 no math, atomic     : %.3f ms\n", run<2>(x, pix2, grid, mx, mn));
  printf("V3 filtered + atomic   : %.3f ms\n", run<3>(x, pix2, grid, mx, mn));
  printf("V4 filtered + dedup    : %.3f ms\n", run<4>(x, pix2, grid, mx, mn));
  printf("V5 fast only + atomic  : %.3f ms\n", run<5>(x, pix2, grid, mx, mn));
  printf("V6 fast only, no atomic: %.3f ms\n", run<6>(x, pix2, grid, mx, mn));
  printf("V7 filtered, no atomic : %.3f ms\n", run<7>(x, pix2, grid, mx, mn));
  printf("V8 fast + compacted at.: %.3f ms\n", run<8>(x, pix2, grid, mx, mn));
  printf("V9 real pix, atomic only: %.3f ms\n", run<9>(x, pix, grid, mx, mn));
  printf("V13 fast + exact dedup   : %.3f ms\n", run<13>(x, pix2, grid, mx, mn));
  printf("V14 filtered + exact ded.: %.3f ms\n", run<14>(x, pix2, grid, mx, mn));
  printf("A10 seq + duplicates     : %.3f ms\n", run<10>(x, pix2, grid, mx, mn));
  printf("A11 ring->row map, arith : %.3f ms\n", run<11>(x, pix2, grid, mx, mn));
  printf("A12 ring->consecutive row: %.3f ms\n", run<12>(x, pix2, grid, mx, mn));
  printf("T0 LDS tile, f64 + flush   : %.3f ms\n", run_tile<0>(x, pix2, grid, mx, mn));
  printf("T1 LDS tile, fast + flush  : %.3f ms\n", run_tile<1>(x, pix2, grid, mx, mn));
  printf("H0 hoisted loads, f64 + atomic : %.3f ms\n", run_hoist<0>(x, pix2, grid, mx, mn));
  printf("H1 hoisted loads, fast + atomic: %.3f ms\n", run_hoist<1>(x, pix2, grid, mx, mn));
  // agreement of the filtered pixel ids with the float64 ones
  std::vector<int> a((size_t)B * N), b((size_t)B * N);
  run<1>(x, pix, grid, mx, mn, 1); run<3>(x, pix2, grid, mx, mn, 1);
  hipMemcpy(a.data(), pix, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), pix2, b.size() * 4, hipMemcpyDeviceToHost);
  size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
  printf("filtered vs f64 pixel ids: %zu differ of %zu\n", diff, a.size());
  return 0;
}
