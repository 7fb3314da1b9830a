// Device-side building blocks shared by the Level-1 and batched kernels (gfx950, wave64).
//
// Everything that decides a pixel id or a depth is IEEE float64 with FMA contraction off
// (the library is built with -ffp-contract=off): the reference bins in float64 and binning in
// float32 moves 11 of 120 000 pixel ids.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/real3daug_hip.h"

#define R3D_SENT 0xFFFFFFFFFFFFFFFFull   // empty pixel: above the bits of every finite double
#define R3D_EMPTY_DEPTH 500.0             // insertion.py:99
#define R3D_WAVE 64

namespace r3d {

constexpr double kPi = 3.14159265358979323846;      // == math.pi == np.pi
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;

struct Sph {
  double r, az, el;
};

// insertion.py:74-76.  x*x + y*y + z*z left to right, no contraction; sqrt and / are the
// correctly rounded IEEE operations; atan2/acos come from the ROCm device library (<= 1-2 ULP,
// as NumPy's own SVML/libm kernels are).
__device__ __forceinline__ Sph spherical(double x, double y, double z) {
  Sph s;
  s.r = sqrt(x * x + y * y + z * z);
  s.az = atan2(y, x) + kPi;
  s.el = acos(z / s.r);
  return s;
}

struct Binning {
  double min_el, max_el, d_el, d_az;
  int rows, cols;
};

__device__ __forceinline__ Binning make_binning(double max_el, double min_el, int rows, int cols) {
  Binning b;
  b.min_el = min_el;
  b.max_el = max_el;
  b.d_el = (max_el - min_el) / (double)rows;   // insertion.py:96
  b.d_az = kTwoPi / (double)cols;              // insertion.py:97
  b.rows = rows;
  b.cols = cols;
  return b;
}

// Python's float % for a positive divisor.
__device__ __forceinline__ double pymod(double a, double m) {
  if (a >= 0.0 && a < m) return a;
  double f = fmod(a, m);
  if (f != 0.0 && f < 0.0) f += m;
  return f;
}

// insertion.py:104-112.  Returns a bit set: 1 = row in range, 2 = column in range.  int()
// truncates toward zero, so (-1, 0) lands in row 0; NaN fails both tests.
__device__ __forceinline__ int bin_point(const Binning &b, double az, double el, int &row, int &col) {
  double tr = trunc((el - b.min_el - 0.00001) / b.d_el);
  double tc = trunc(pymod(az, kTwoPi) / b.d_az);
  int ok = 0;
  row = col = 0;
  if (tr >= 0.0 && tr < (double)b.rows) {
    ok |= 1;
    row = (int)tr;
  }
  if (tc >= 0.0 && tc < (double)b.cols) {
    ok |= 2;
    col = (int)tc;
  }
  return ok;
}

__device__ __forceinline__ unsigned long long depth_key(double r) {
  return (unsigned long long)__double_as_longlong(r);
}
__device__ __forceinline__ double key_depth(unsigned long long k) {
  return __longlong_as_double((long long)k);
}

// Order-preserving map double -> u64 for values of either sign (used for z/r reductions).
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_key_inv(unsigned long long k) {
  unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}

// ---- wave / block reductions and scans (wave = 64 lanes) ------------------------------------
// Cross-lane steps are DPP row shifts and row broadcasts (VALU, a few cycles each), not ds_bpermute
// (an LDS-pipe round trip per step): lane i takes lane i-1, i-2, i-4, i-8 of its row of 16, then
// lane 15 of the previous row, then lane 31.  Lanes without a source keep the identity, so after the
// six steps lane i holds the combination of lanes 0..i (an inclusive scan) and lane 63 the total.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_take(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false);
}
#define R3D_DPP_STEPS(STEP) \
  STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)

__device__ __forceinline__ int wave_last_i32(int v) { return __builtin_amdgcn_readlane(v, 63); }

// Inclusive wave scan of ints.
__device__ __forceinline__ int wave_iscan_i32(int v) {
#define R3D_STEP(C, M) v += dpp_take<C, M>(0, v);
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) { return wave_last_i32(wave_iscan_i32(v)); }
__device__ __forceinline__ int wave_or_i32(int v) {
#define R3D_STEP(C, M) v |= dpp_take<C, M>(0, v);
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return wave_last_i32(v);
}
__device__ __forceinline__ int wave_min_i32(int v) {
#define R3D_STEP(C, M) { int t = dpp_take<C, M>(0x7FFFFFFF, v); v = t < v ? t : v; }
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return wave_last_i32(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#define R3D_STEP(C, M) { int t = dpp_take<C, M>((int)0x80000000, v); v = t > v ? t : v; }
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return wave_last_i32(v);
}
__device__ __forceinline__ float wave_min_f32(float v) {           // NaN-free inputs
#define R3D_STEP(C, M) v = fminf(v, __int_as_float(dpp_take<C, M>(0x7F800000, __float_as_int(v))));
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return __int_as_float(wave_last_i32(__float_as_int(v)));
}
__device__ __forceinline__ float wave_max_f32(float v) {
#define R3D_STEP(C, M) v = fmaxf(v, __int_as_float(dpp_take<C, M>((int)0xFF800000, __float_as_int(v))));
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return __int_as_float(wave_last_i32(__float_as_int(v)));
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#define R3D_STEP(C, M)                                                                                    \
  {                                                                                                       \
    unsigned int lo = (unsigned int)dpp_take<C, M>(-1, (int)(unsigned int)v);                             \
    unsigned int hi = (unsigned int)dpp_take<C, M>(-1, (int)(unsigned int)(v >> 32));                     \
    unsigned long long t = ((unsigned long long)hi << 32) | lo;                                           \
    v = t < v ? t : v;                                                                                    \
  }
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return ((unsigned long long)(unsigned int)wave_last_i32((int)(unsigned int)(v >> 32)) << 32) |
         (unsigned int)wave_last_i32((int)(unsigned int)v);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define R3D_STEP(C, M)                                                                                    \
  {                                                                                                       \
    unsigned int lo = (unsigned int)dpp_take<C, M>(0, (int)(unsigned int)v);                              \
    unsigned int hi = (unsigned int)dpp_take<C, M>(0, (int)(unsigned int)(v >> 32));                      \
    unsigned long long t = ((unsigned long long)hi << 32) | lo;                                           \
    v = t > v ? t : v;                                                                                    \
  }
  R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
  return ((unsigned long long)(unsigned int)wave_last_i32((int)(unsigned int)(v >> 32)) << 32) |
         (unsigned int)wave_last_i32((int)(unsigned int)v);
}

// Exclusive block scan; `sm` needs blockDim.x/64 + 1 ints.  Returns the exclusive prefix of v and
// the block total in `total`.  Contains two barriers; every thread of the block must call it.
__device__ __forceinline__ int block_escan_i32(int v, int *sm, int &total) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  int inc = wave_iscan_i32(v);
  if (lane == 63) sm[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int t = lane < nw ? sm[lane] : 0;
    int ti = wave_iscan_i32(t);
    if (lane < nw) sm[lane] = ti - t;
    if (lane == nw - 1) sm[nw] = ti;
  }
  __syncthreads();
  int base = sm[wave];
  total = sm[nw];
  __syncthreads();
  return base + inc - v;
}

}  // namespace r3d
