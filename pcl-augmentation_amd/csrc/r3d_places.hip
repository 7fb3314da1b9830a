// Level 3: the placement search for gfx950.
//
// Reference: find_possible_places, semantic_segmentation/Real3DAug/tools/find_spot.py:192-273, with
// rotate_bounding_box_2 (:42-76), check_bounding_box (:79-104), correct_height (:107-152) and
// cut_bounding_box (tools/cut_bbox.py:7-68).  The reference walks the 360 one-degree steps one
// after the other and scans the whole scene several times per step.  Here the work is regrouped
// by what it depends on:
//
//   k_place_chain        per query, sequential: the box centre and orientation of every step depend
//                        only on the sample's annotation (scipy Rotation round trips + BLAS products,
//                        restated with explicit fma() in the order the BLAS kernels use).
//   k_place_road_min     per original point: a point can only matter for the steps whose box centre
//   k_place_surface_gather  is within the search radius, i.e. for a short arc of steps; it updates the
//                        minimum distance of those steps (LDS atomics, flushed once per block), and in
//                        the second pass appends itself to the steps whose first non-empty radius
//                        contains it.  The wide 5 m search only runs for steps a 0.6 m search left open.
//   k_place_road_level   per step: the surface points in the reference's order (label order of the
//                        config, then point order), summed one by one like np.mean over rows does.
//   k_place_scene_in_box per current-cloud point that is not placement surface: six-plane test
//                        against the box of the steps on its arc.
//   k_place_sample_chain per query, one workgroup: the sample's points live in registers and take
//                        the 360 rotations, map tests, height corrections and scene-box tests in
//                        order; possible placements are written out as they are found.
//
// Everything that decides an outcome is float64 in the reference's operation order (the library is
// built with -ffp-contract=off; the fused operations below are the ones the BLAS performs).
#include <cstring>

#include "r3d_device.hpp"
#include "r3d_host.hpp"

namespace {
using namespace r3d;

constexpr int kRot = R3D_PLACE_ROTATIONS;
constexpr int kCap = R3D_PLACE_SURFACE_CAP;
constexpr int kPB = 256;                 // threads per block
constexpr int kPointsPerBlock = 4096;    // points of one query handled by one block of the point passes
constexpr int kBoxD = 15;                // 3x3 matrix, upper planes, lower planes
constexpr double kCos1 = 0x1.ffec097f5af8ap-1;   // np.cos(np.deg2rad(1)), find_spot.py:52-59
constexpr double kSin1 = 0x1.1df0b2b89dd1ep-6;   // np.sin(np.deg2rad(1))
constexpr float kDegPerRad = 57.29577951308232f;

struct PlaceWs {
  double *cx, *cy;              // [Q][360] box centre after step r
  double *quat;                 // [Q][360][4] box orientation after step r
  double *rotm;                 // [Q][360][9] its matrix as cut_bounding_box builds it
  unsigned long long *dmin;     // [Q][360] min squared distance of a surface point to the centre
  int32_t *kstar;               // [Q][360] index of the first search radius that holds surface, -1 none
  double *road;                 // [Q][360] mean surface height
  double *planes;               // [Q][360][6] the sample box's upper / lower planes per axis
  int32_t *surf_n;              // [Q][360]
  unsigned long long *surf;     // [Q][360][kCap] (label rank << 40 | point index)
  uint32_t *hit;                // [Q][12] bit r: a non-surface scene point is inside the box of step r
  unsigned long long *gather_sq;// [Q] largest squared radius any step of the query needs (bits of a double)
  double *boxes;                // [Q][max_boxes][kBoxD]
  size_t total;
};

PlaceWs carve_places(int32_t nq, int32_t max_boxes, void *base) {
  Carver c(base);
  PlaceWs w;
  size_t qr = (size_t)nq * kRot;
  w.cx = c.take<double>(qr);
  w.cy = c.take<double>(qr);
  w.quat = c.take<double>(qr * 4);
  w.rotm = c.take<double>(qr * 9);
  w.dmin = c.take<unsigned long long>(qr);
  w.kstar = c.take<int32_t>(qr);
  w.road = c.take<double>(qr);
  w.planes = c.take<double>(qr * 6);
  w.surf_n = c.take<int32_t>(qr);
  w.surf = c.take<unsigned long long>(qr * kCap);
  w.hit = c.take<uint32_t>((size_t)nq * 12);
  w.gather_sq = c.take<unsigned long long>((size_t)nq);
  w.boxes = c.take<double>((size_t)nq * (max_boxes > 0 ? max_boxes : 1) * kBoxD);
  w.total = c.off;
  return w;
}

struct Radii {
  double sq[R3D_PLACE_MAX_RADII];
  int n;
};

// ---- scipy.spatial.transform.Rotation, the calls on the path (SciPy 1.15: from_quat normalises,
// as_matrix, from_matrix by the largest of diagonal and trace) --------------------------------
struct Quat {
  double x, y, z, w;
};

__device__ __forceinline__ Quat quat_normalize(Quat q) {
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}

__device__ __forceinline__ void quat_to_matrix(Quat q, double (&m)[9]) {
  double x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
  double xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
  m[0] = x2 - y2 - z2 + w2;
  m[1] = 2 * (xy - zw);
  m[2] = 2 * (xz + yw);
  m[3] = 2 * (xy + zw);
  m[4] = -x2 + y2 - z2 + w2;
  m[5] = 2 * (yz - xw);
  m[6] = 2 * (xz - yw);
  m[7] = 2 * (yz + xw);
  m[8] = -x2 - y2 + z2 + w2;
}

__device__ __forceinline__ Quat matrix_to_quat(const double (&m)[9]) {
  double d3 = m[0] + m[4] + m[8];
  int c = 0;
  double best = m[0];
  if (m[4] > best) { best = m[4]; c = 1; }
  if (m[8] > best) { best = m[8]; c = 2; }
  if (d3 > best) c = 3;
  double q[4];
  if (c != 3) {
    int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1 - d3 + 2 * m[i * 3 + i];
    q[j] = m[j * 3 + i] + m[i * 3 + j];
    q[k] = m[k * 3 + i] + m[i * 3 + k];
    q[3] = m[k * 3 + j] - m[j * 3 + k];
  } else {
    q[0] = m[7] - m[5];
    q[1] = m[2] - m[6];
    q[2] = m[3] - m[1];
    q[3] = 1 + d3;
  }
  return quat_normalize(Quat{q[0], q[1], q[2], q[3]});
}

// tools/cut_bbox.py:30-64: per box axis a the column (R[0][a], R[1][a], R[2][a]) and the two plane
// offsets; the box spans +-length/2, +-width/2 and 0..height from its (bottom) centre.
__device__ __forceinline__ void box_planes(const double (&R)[9], double xc, double yc, double zc, double length,
                                           double width, double height, double (&up)[3], double (&dn)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double r0 = R[a], r1 = R[3 + a], r2 = R[6 + a];
    if (a < 2) {
      double e = a == 0 ? length : width;
      up[a] = r0 * (xc + r0 * e / 2) + r1 * (yc + r1 * e / 2) + r2 * (zc + r2 * e / 2);
      dn[a] = r0 * (xc - r0 * e / 2) + r1 * (yc - r1 * e / 2) + r2 * (zc - r2 * e / 2);
    } else {
      up[a] = r0 * (xc + r0 * height) + r1 * (yc + r1 * height) + r2 * (zc + r2 * height);
      dn[a] = r0 * (xc - r0 * 0.0) + r1 * (yc - r1 * 0.0) + r2 * (zc - r2 * 0.0);
    }
  }
}

__device__ __forceinline__ bool inside_box(const double *R, const double *up, const double *dn, double x, double y,
                                           double z) {
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double lhs = R[a] * x + R[3 + a] * y + R[6 + a] * z;
    in = in && (lhs < up[a]) && (lhs > dn[a]);
  }
  return in;
}

// ---- k_place_chain: find_spot.py:52-70 applied 360 times to the annotation -----------------------
__global__ void k_place_chain(const r3d_place_query_t *Q, int nq, PlaceWs w, int32_t *status) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const r3d_place_query_t &qq = Q[q];
  double c0 = qq.anno[0], c1 = qq.anno[1], c2 = qq.anno[2];
  Quat a{qq.anno[3], qq.anno[4], qq.anno[5], qq.anno[6]};
  bool finite = true;
  for (int i = 0; i < 10; ++i) finite = finite && isfinite(qq.anno[i]);
  for (int i = 0; i < 8; ++i) finite = finite && isfinite(qq.pose[i]);
  status[q] = finite ? 0 : R3D_PS_NONFINITE;
  w.gather_sq[q] = 0ull;
  const double Z[9] = {kCos1, -kSin1, 0.0, kSin1, kCos1, 0.0, 0.0, 0.0, 1.0};
  Quat n = quat_normalize(a);                                   // R.from_quat(annotation[1]), :53
  for (int r = 0; r < kRot; ++r) {
    double Rm[9], F[9];
    quat_to_matrix(n, Rm);                                      // :55
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)                               // np.dot(rot_matrix, z_rot_matrix), :61
        F[i * 3 + j] = fma(Rm[i * 3 + 2], Z[6 + j], fma(Rm[i * 3 + 1], Z[3 + j], Rm[i * 3 + 0] * Z[j]));
    a = matrix_to_quat(F);                                      // :63-65
    double n0 = fma(Z[2], c2, fma(Z[0], c0, Z[1] * c1));        // np.dot(z_rot_matrix, position), :66-70
    double n1 = fma(Z[5], c2, fma(Z[3], c0, Z[4] * c1));
    double n2 = fma(Z[8], c2, fma(Z[6], c0, Z[7] * c1));
    c0 = n0;
    c1 = n1;
    c2 = n2;
    size_t o = (size_t)q * kRot + r;
    w.cx[o] = c0;
    w.cy[o] = c1;
    w.quat[o * 4 + 0] = a.x;
    w.quat[o * 4 + 1] = a.y;
    w.quat[o * 4 + 2] = a.z;
    w.quat[o * 4 + 3] = a.w;
    n = quat_normalize(a);                                      // from_quat of the next step and of cut_bbox.py:26
    double Rb[9];
    quat_to_matrix(n, Rb);
    for (int i = 0; i < 9; ++i) w.rotm[o * 9 + i] = Rb[i];
  }
}

// One thread per annotated scene box of a query (scene_annotation[i], find_spot.py:99-101).
__global__ void k_place_boxes(const r3d_place_query_t *Q, int nq, int max_boxes, PlaceWs w) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * max_boxes) return;
  int q = t / max_boxes, j = t % max_boxes;
  const r3d_place_query_t &qq = Q[q];
  if (j >= qq.n_boxes) return;
  const double *b = qq.boxes + (size_t)j * 10;
  double R[9], up[3], dn[3];
  quat_to_matrix(quat_normalize(Quat{b[3], b[4], b[5], b[6]}), R);
  box_planes(R, b[0], b[1], b[2], b[7], b[8], b[9], up, dn);
  double *o = w.boxes + ((size_t)q * max_boxes + j) * kBoxD;
  for (int i = 0; i < 9; ++i) o[i] = R[i];
  for (int i = 0; i < 3; ++i) {
    o[9 + i] = up[i];
    o[12 + i] = dn[i];
  }
}

// The steps whose box centre can be within `reach` of the point (x, y): the centres lie on a circle
// around the sensor, one degree apart.  Float32 and generous: +-1.5 steps and `reach` padded by the
// caller.  first = index (step - 1) of the first candidate, count of candidates (<= 360).
__device__ __forceinline__ bool steps_in_reach(float x, float y, float rho_c, float th1, float reach, int &first,
                                               int &count) {
  float rho = sqrtf(x * x + y * y);
  if (!(fabsf(rho - rho_c) <= reach)) return false;
  first = 0;
  count = kRot;
  float den = 2.f * rho * rho_c;
  if (den < 1e-6f) return true;
  float t = (rho * rho + rho_c * rho_c - reach * reach) / den;
  if (t <= -1.f) return true;
  float half = (t >= 1.f ? 0.f : acosf(t)) * kDegPerRad + 1.5f;
  float rel = (atan2f(y, x) - th1) * kDegPerRad;
  int lo = (int)floorf(rel - half), hi = (int)ceilf(rel + half);
  if (hi - lo + 1 >= kRot) return true;
  count = hi - lo + 1;
  first = ((lo % kRot) + kRot) % kRot;
  return true;
}

__device__ __forceinline__ int label_rank(const r3d_place_query_t &qq, double label) {
  for (int j = 0; j < qq.n_ok_labels; ++j)
    if (label == (double)qq.ok_labels[j]) return j;
  return -1;
}

// ---- k_place_road_min: correct_height's distance test (find_spot.py:123) for every step at once ---
// mode 0: search within `reach` for every step.  mode 1: only the steps whose minimum is still above
// resolved_sq (the narrow pass found nothing that close, so it may have missed the true minimum).
__global__ __launch_bounds__(kPB) void k_place_road_min(const r3d_place_query_t *Q, PlaceWs w, float reach, int mode,
                                                        double resolved_sq) {
  const int q = blockIdx.y, tid = threadIdx.x;
  const r3d_place_query_t &qq = Q[q];
  const int64_t n = qq.n_orig, start = (int64_t)blockIdx.x * kPointsPerBlock;
  if (start >= n) return;
  __shared__ double s_cx[kRot], s_cy[kRot];
  __shared__ unsigned long long s_min[kRot];
  __shared__ unsigned char s_need[kRot];
  int any = 0;
  for (int r = tid; r < kRot; r += kPB) {
    size_t o = (size_t)q * kRot + r;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
    s_min[r] = R3D_SENT;
    int need = mode == 0 ? 1 : (w.dmin[o] > depth_key(resolved_sq) ? 1 : 0);
    s_need[r] = (unsigned char)need;
    any |= need;
  }
  if (!__syncthreads_or(any)) return;
  const float rho_c = sqrtf((float)(s_cx[0] * s_cx[0] + s_cy[0] * s_cy[0])), th1 = atan2f((float)s_cy[0], (float)s_cx[0]);
  const int64_t end = start + kPointsPerBlock < n ? start + kPointsPerBlock : n;
  for (int64_t i = start + tid; i < end; i += kPB) {
    const double *p = qq.orig + i * qq.orig_ld;
    double x = p[0], y = p[1], z = p[2];
    if (!(z > -3.0)) continue;                                    // :133-134
    if (label_rank(qq, p[qq.orig_label_col]) < 0) continue;       // :125-131
    int first, count;
    if (!steps_in_reach((float)x, (float)y, rho_c, th1, reach, first, count)) continue;
    for (int t = 0; t < count; ++t) {
      int r = first + t;
      r = r >= kRot ? r - kRot : r;
      if (!s_need[r]) continue;
      double dx = x - s_cx[r], dy = y - s_cy[r];
      unsigned long long key = depth_key(dx * dx + dy * dy);     // :123, non-negative: bits are ordered
      if (key < s_min[r]) atomicMin(&s_min[r], key);
    }
  }
  __syncthreads();
  for (int r = tid; r < kRot; r += kPB)
    if (s_min[r] != R3D_SENT) atomicMin(&w.dmin[(size_t)q * kRot + r], s_min[r]);
}

// First radius of the growing search that holds surface (find_spot.py:121-140).
__global__ void k_place_kstar(int nq, PlaceWs w, Radii rad) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * kRot) return;
  unsigned long long key = w.dmin[t];
  int k = -1;
  if (key != R3D_SENT) {
    double d2 = key_depth(key);
    for (int j = 0; j < rad.n; ++j)
      if (d2 <= rad.sq[j]) {
        k = j;
        break;
      }
  }
  w.kstar[t] = k;
  w.surf_n[t] = 0;
  if (k >= 0) atomicMax(&w.gather_sq[t / kRot], depth_key(rad.sq[k]));
}

// ---- k_place_surface_gather: the points of `surface` (find_spot.py:123-134) of every step ---------
__global__ __launch_bounds__(kPB) void k_place_surface_gather(const r3d_place_query_t *Q, PlaceWs w, Radii rad) {
  const int q = blockIdx.y, tid = threadIdx.x;
  const r3d_place_query_t &qq = Q[q];
  const int64_t n = qq.n_orig, start = (int64_t)blockIdx.x * kPointsPerBlock;
  if (start >= n) return;
  const double reach_sq = key_depth(w.gather_sq[q]);
  if (!(reach_sq > 0.0)) return;                                  // no step found surface
  __shared__ double s_cx[kRot], s_cy[kRot], s_thr[kRot];
  for (int r = tid; r < kRot; r += kPB) {
    size_t o = (size_t)q * kRot + r;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
    int k = w.kstar[o];
    s_thr[r] = k >= 0 ? rad.sq[k] : -1.0;
  }
  __syncthreads();
  const float reach = (float)sqrt(reach_sq) * 1.01f + 0.05f;
  const float rho_c = sqrtf((float)(s_cx[0] * s_cx[0] + s_cy[0] * s_cy[0])), th1 = atan2f((float)s_cy[0], (float)s_cx[0]);
  const int64_t end = start + kPointsPerBlock < n ? start + kPointsPerBlock : n;
  for (int64_t i = start + tid; i < end; i += kPB) {
    const double *p = qq.orig + i * qq.orig_ld;
    double x = p[0], y = p[1], z = p[2];
    if (!(z > -3.0)) continue;
    int rank = label_rank(qq, p[qq.orig_label_col]);
    if (rank < 0) continue;
    int first, count;
    if (!steps_in_reach((float)x, (float)y, rho_c, th1, reach, first, count)) continue;
    for (int t = 0; t < count; ++t) {
      int r = first + t;
      r = r >= kRot ? r - kRot : r;
      double dx = x - s_cx[r], dy = y - s_cy[r];
      if (!(dx * dx + dy * dy <= s_thr[r])) continue;             // :123 with the step's radius
      size_t o = (size_t)q * kRot + r;
      int slot = atomicAdd(&w.surf_n[o], 1);
      if (slot < kCap) w.surf[o * kCap + slot] = ((unsigned long long)rank << 40) | (unsigned long long)i;
    }
  }
}

// ---- k_place_road_level: np.mean(surface, axis=0)[2] (find_spot.py:144) and the box planes --------
__global__ void k_place_road_level(const r3d_place_query_t *Q, int nq, PlaceWs w, int32_t *status) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * kRot) return;
  const int q = t / kRot;
  const r3d_place_query_t &qq = Q[q];
  double road = 0.0;
  if (w.kstar[t] >= 0) {
    int n = w.surf_n[t];
    if (n > kCap) {
      atomicOr(&status[q], R3D_PS_SURFACE_OVERFLOW);
      n = kCap;
    }
    unsigned long long *l = w.surf + (size_t)t * kCap;
    for (int i = 1; i < n; ++i) {                                 // label order of the config, then point order
      unsigned long long v = l[i];
      int j = i - 1;
      while (j >= 0 && l[j] > v) {
        l[j + 1] = l[j];
        --j;
      }
      l[j + 1] = v;
    }
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += qq.orig[(int64_t)(l[i] & ((1ull << 40) - 1)) * qq.orig_ld + 2];
    road = acc / (double)n;
  }
  w.road[t] = road;
  double R[9], up[3], dn[3];
  for (int i = 0; i < 9; ++i) R[i] = w.rotm[(size_t)t * 9 + i];
  box_planes(R, w.cx[t], w.cy[t], road, qq.anno[7], qq.anno[8], qq.anno[9], up, dn);
  for (int i = 0; i < 3; ++i) {
    w.planes[(size_t)t * 6 + i] = up[i];
    w.planes[(size_t)t * 6 + 3 + i] = dn[i];
  }
}

// ---- k_place_scene_in_box: cut_bounding_box(scene_pcl, sample_anno) minus surface (:91-97) --------
__global__ __launch_bounds__(kPB) void k_place_scene_in_box(const r3d_place_query_t *Q, PlaceWs w) {
  const int q = blockIdx.y, tid = threadIdx.x;
  const r3d_place_query_t &qq = Q[q];
  const int64_t n = qq.n_scene, start = (int64_t)blockIdx.x * kPointsPerBlock;
  if (start >= n) return;
  __shared__ double s_cx[kRot], s_cy[kRot];
  __shared__ unsigned char s_near[kRot];
  __shared__ uint32_t s_hit[12];
  int any = 0;
  for (int r = tid; r < kRot; r += kPB) {
    size_t o = (size_t)q * kRot + r;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
    int near = w.kstar[o] >= 0;
    s_near[r] = (unsigned char)near;
    any |= near;
  }
  if (tid < 12) s_hit[tid] = 0u;
  if (!__syncthreads_or(any)) return;
  const double l = qq.anno[7], wd = qq.anno[8], h = qq.anno[9];
  const double reach_d = sqrt(l * l / 4 + wd * wd / 4 + h * h);     // no box point is further from the centre
  const float reach = (float)reach_d * 1.01f + 0.05f;
  const double reach_sq = (double)reach * (double)reach;
  const float rho_c = sqrtf((float)(s_cx[0] * s_cx[0] + s_cy[0] * s_cy[0])), th1 = atan2f((float)s_cy[0], (float)s_cx[0]);
  const int64_t end = start + kPointsPerBlock < n ? start + kPointsPerBlock : n;
  for (int64_t i = start + tid; i < end; i += kPB) {
    const double *p = qq.scene + i * qq.scene_ld;
    double x = p[0], y = p[1], z = p[2];
    if (label_rank(qq, p[qq.scene_label_col]) >= 0) continue;     // :94-95: surface may be inside the box
    int first, count;
    if (!steps_in_reach((float)x, (float)y, rho_c, th1, reach, first, count)) continue;
    for (int t = 0; t < count; ++t) {
      int r = first + t;
      r = r >= kRot ? r - kRot : r;
      if (!s_near[r]) continue;
      double dx = x - s_cx[r], dy = y - s_cy[r];
      if (dx * dx + dy * dy > reach_sq) continue;
      size_t o = (size_t)q * kRot + r;
      if (inside_box(w.rotm + o * 9, w.planes + o * 6, w.planes + o * 6 + 3, x, y, z))
        atomicOr(&s_hit[r >> 5], 1u << (r & 31));
    }
  }
  __syncthreads();
  if (tid < 12 && s_hit[tid]) atomicOr(&w.hit[(size_t)q * 12 + tid], s_hit[tid]);
}

// ---- k_place_sample_chain: the loop of find_spot.py:228-269 on the sample's points ---------------
template <int PPT>
__global__ __launch_bounds__(kPB) void k_place_sample_chain(const r3d_place_query_t *Q, PlaceWs w, int max_boxes,
                                                           uint8_t *flags, int32_t *n_possible, int32_t *rot_out,
                                                           double *anno_out, double *cand, int32_t first_cand,
                                                           int32_t *status) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const r3d_place_query_t &qq = Q[q];
  const int m = qq.m;
  double x[PPT], y[PPT], z[PPT];
  int bad_input = 0;
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    int i = tid + u * kPB;
    x[u] = y[u] = z[u] = 0.0;
    if (i < m) {
      x[u] = qq.sample[(size_t)i * 5 + 0];
      y[u] = qq.sample[(size_t)i * 5 + 1];
      z[u] = qq.sample[(size_t)i * 5 + 2];
      if (!(isfinite(x[u]) && isfinite(y[u]) && isfinite(z[u]))) bad_input = 1;
    }
  }
  if (bad_input) atomicOr(&status[q], R3D_PS_NONFINITE);
  const double T00 = qq.pose[0], T01 = qq.pose[1], T02 = qq.pose[2], T03 = qq.pose[3];
  const double T10 = qq.pose[4], T11 = qq.pose[5], T12 = qq.pose[6], T13 = qq.pose[7];
  const double mv0 = qq.map_move[0], mv1 = qq.map_move[1];
  const long long rows = qq.map_rows, cols = qq.map_cols;
  const int nb = qq.n_boxes;
  const double *boxes = w.boxes + (size_t)q * max_boxes * kBoxD;
  double anno_z = qq.anno[2];
  int n_out = 0;
  for (int r = 0; r < kRot; ++r) {
    const size_t o = (size_t)q * kRot + r;
    int bad = 0;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      if (tid + u * kPB >= m) continue;
      // bbox_pcl[:, :3] = (z_rot_matrix @ bbox_pcl[:, :3].T).T, :72
      double nx = fma(0.0, z[u], fma(-kSin1, y[u], kCos1 * x[u]));
      double ny = fma(0.0, z[u], fma(kCos1, y[u], kSin1 * x[u]));
      double nz = fma(1.0, z[u], fma(0.0, y[u], 0.0 * x[u]));
      x[u] = nx;
      y[u] = ny;
      z[u] = nz;
      // transformation_matrix @ [x y z 1], minus map_move, astype(int): :234-238
      double g0 = fma(T03, 1.0, fma(T02, nz, fma(T01, ny, T00 * nx))) - mv0;
      double g1 = fma(T13, 1.0, fma(T12, nz, fma(T11, ny, T10 * nx))) - mv1;
      long long i0 = (long long)g0, i1 = (long long)g1;
      if (i0 < rows && i0 > -1 && i1 < cols && i1 > -1) {         // :240-243
        unsigned v = qq.map[i0 * cols + i1];
        if (!((qq.ok_map[v >> 6] >> (v & 63)) & 1ull)) bad = 1;   // :245-248
      }
    }
    const bool on_surface = !__syncthreads_or(bad);
    const bool near = w.kstar[o] >= 0;
    if (on_surface && near) {                                     // correct_height, :142-148
      double road = w.road[o];
      double z_move = road - anno_z;
#pragma unroll
      for (int u = 0; u < PPT; ++u) z[u] += z_move;
      anno_z = road;
    }
    const bool scene_hit = (w.hit[(size_t)q * 12 + (r >> 5)] >> (r & 31)) & 1u;
    bool sample_hit = false;
    if (on_surface && near && !scene_hit && nb > 0) {             // :99-103
      int in_any = 0;
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        if (tid + u * kPB >= m) continue;
        for (int b = 0; b < nb; ++b) {
          const double *bx = boxes + (size_t)b * kBoxD;
          if (inside_box(bx, bx + 9, bx + 12, x[u], y[u], z[u])) in_any = 1;
        }
      }
      sample_hit = __syncthreads_or(in_any);
    }
    const bool possible = on_surface && near && !scene_hit && !sample_hit;
    if (tid == 0)
      flags[o] = (uint8_t)((on_surface ? R3D_PF_ON_SURFACE : 0) | (near ? R3D_PF_NEAR_ROAD : 0) |
                           (scene_hit ? R3D_PF_SCENE_IN_BOX : 0) | (sample_hit ? R3D_PF_SAMPLE_IN_BOX : 0) |
                           (possible ? R3D_PF_POSSIBLE : 0));
    if (possible) {                                               // :257-264
      int j = n_out - first_cand;
      if (j >= 0 && j < qq.cand_cap) {
        double *out = cand + qq.cand_off + (size_t)j * m * 5;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
          int i = tid + u * kPB;
          if (i >= m) continue;
          out[(size_t)i * 5 + 0] = x[u];
          out[(size_t)i * 5 + 1] = y[u];
          out[(size_t)i * 5 + 2] = z[u];
          out[(size_t)i * 5 + 3] = qq.sample[(size_t)i * 5 + 3];
          out[(size_t)i * 5 + 4] = qq.sample[(size_t)i * 5 + 4];
        }
      }
      if (tid == 0) {
        size_t oo = (size_t)q * kRot + n_out;
        rot_out[oo] = r + 1;
        anno_out[oo * 7 + 0] = w.cx[o];
        anno_out[oo * 7 + 1] = w.cy[o];
        anno_out[oo * 7 + 2] = anno_z;
        for (int i = 0; i < 4; ++i) anno_out[oo * 7 + 3 + i] = w.quat[o * 4 + i];
      }
      ++n_out;
    }
  }
  if (tid == 0) n_possible[q] = n_out;
}

}  // namespace

extern "C" size_t r3d_places_workspace_bytes(int32_t n_queries, int32_t max_boxes) {
  if (n_queries <= 0 || max_boxes < 0) return 0;
  return carve_places(n_queries, max_boxes, nullptr).total;
}

extern "C" int r3d_find_possible_places(const r3d_place_query_t *queries, int32_t n_queries, int64_t max_n_scene,
                                        int64_t max_n_orig, int32_t max_m, int32_t max_boxes,
                                        const double *radius_sq, int32_t n_radii, uint8_t *flags,
                                        int32_t *n_possible, int32_t *rot_out, double *anno_out, double *cand,
                                        int32_t first_cand, int32_t *status, void *workspace,
                                        size_t workspace_bytes, void *stream) {
  if (!queries || !radius_sq || !flags || !n_possible || !rot_out || !anno_out || !cand || !status || !workspace)
    return fail(R3D_E_ARG, "places: null argument");
  if (n_queries <= 0 || n_queries > 65535) return fail(R3D_E_ARG, "places: 1..65535 queries per call");
  if (max_n_scene < 0 || max_n_orig < 0 || max_m <= 0 || max_boxes < 0 || first_cand < 0)
    return fail(R3D_E_ARG, "places: negative size");
  if (max_m > kPB * 32) return fail(R3D_E_ARG, "places: samples are limited to 8192 points");
  if (n_radii <= 0 || n_radii > R3D_PLACE_MAX_RADII) return fail(R3D_E_ARG, "places: 1..64 search radii");
  PlaceWs w = carve_places(n_queries, max_boxes, workspace);
  if (workspace_bytes < w.total) return fail(R3D_E_WORKSPACE, "places: workspace smaller than r3d_places_workspace_bytes()");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Radii rad;
  memset(&rad, 0, sizeof rad);
  rad.n = n_radii;
  double reach_max = 0.0;
  for (int i = 0; i < n_radii; ++i) {
    rad.sq[i] = radius_sq[i];
    if (!(radius_sq[i] > 0.0)) return fail(R3D_E_ARG, "places: radii must be positive");
    if (radius_sq[i] > reach_max) reach_max = radius_sq[i];
  }
  const size_t qr = (size_t)n_queries * kRot;
  R3D_HIP(hipMemsetAsync(w.dmin, 0xFF, qr * sizeof(unsigned long long), st));
  R3D_HIP(hipMemsetAsync(w.hit, 0, (size_t)n_queries * 12 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_place_chain, dim3((n_queries + 63) / 64), dim3(64), 0, st, queries, n_queries, w, status);
  if (max_boxes > 0)
    hipLaunchKernelGGL(k_place_boxes, dim3((n_queries * max_boxes + 255) / 256), dim3(256), 0, st, queries,
                       n_queries, max_boxes, w);
  const int pb_orig = (int)((max_n_orig + kPointsPerBlock - 1) / kPointsPerBlock);
  const int pb_scene = (int)((max_n_scene + kPointsPerBlock - 1) / kPointsPerBlock);
  const double narrow = 0.6;     // metres: what the first distance pass looks at
  if (pb_orig > 0) {
    hipLaunchKernelGGL(k_place_road_min, dim3(pb_orig, n_queries), dim3(kPB), 0, st, queries, w,
                       (float)(narrow * 1.01 + 0.05), 0, 0.0);
    if (reach_max > narrow * narrow)
      hipLaunchKernelGGL(k_place_road_min, dim3(pb_orig, n_queries), dim3(kPB), 0, st, queries, w,
                         (float)(sqrt(reach_max) * 1.01 + 0.05), 1, narrow * narrow);
  }
  hipLaunchKernelGGL(k_place_kstar, dim3((unsigned)((qr + 255) / 256)), dim3(256), 0, st, n_queries, w, rad);
  if (pb_orig > 0)
    hipLaunchKernelGGL(k_place_surface_gather, dim3(pb_orig, n_queries), dim3(kPB), 0, st, queries, w, rad);
  hipLaunchKernelGGL(k_place_road_level, dim3((unsigned)((qr + 255) / 256)), dim3(256), 0, st, queries, n_queries, w,
                     status);
  if (pb_scene > 0)
    hipLaunchKernelGGL(k_place_scene_in_box, dim3(pb_scene, n_queries), dim3(kPB), 0, st, queries, w);
  const int ppt = (max_m + kPB - 1) / kPB;
#define R3D_CHAIN(P)                                                                                              \
  hipLaunchKernelGGL(k_place_sample_chain<P>, dim3(n_queries), dim3(kPB), 0, st, queries, w, max_boxes, flags,   \
                     n_possible, rot_out, anno_out, cand, first_cand, status)
  if (ppt <= 1) R3D_CHAIN(1);
  else if (ppt <= 2) R3D_CHAIN(2);
  else if (ppt <= 4) R3D_CHAIN(4);
  else if (ppt <= 8) R3D_CHAIN(8);
  else if (ppt <= 16) R3D_CHAIN(16);
  else R3D_CHAIN(32);
#undef R3D_CHAIN
  R3D_LAUNCHED("placement kernels");
  return R3D_OK;
}
