// Level 3: the placement search for gfx950.
//
// Reference: find_possible_places, semantic_segmentation/Real3DAug/tools/find_spot.py:192-273, with
// rotate_bounding_box_2 (:42-76), check_bounding_box (:79-104), correct_height (:107-152) and
// cut_bounding_box (tools/cut_bbox.py:7-68).  The reference walks the 360 one-degree steps one
// after the other and scans the whole scene several times per step.  Here the work is regrouped
// by what it depends on:
//
//   k_place_centres / k_place_orient  per query, sequential: the box centre and orientation of every step depend
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
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "r3d_device.hpp"
#include "r3d_host.hpp"

namespace {
using namespace r3d;

constexpr int kRot = R3D_PLACE_ROTATIONS;
constexpr int kCap = R3D_PLACE_SURFACE_CAP;
constexpr int kPB = 256;                 // threads per block
constexpr int kCB = 512;                 // threads per block of the sample chain
#ifndef R3D_PLACE_BLOCK
#define R3D_PLACE_BLOCK 4096
#endif
#ifndef R3D_PLACE_TURNS
#define R3D_PLACE_TURNS 1
#endif
constexpr int kPointsPerBlock = R3D_PLACE_BLOCK;    // points of one query per turn of a block of the point passes (one ballot lists its chunks)
constexpr int kTurns = R3D_PLACE_TURNS;                // turns per block: a block stages its query's tables (centres, minima, boxes of the 360
                                         // steps: 9-25 KB) once for 16 384 points -- at 320 queries the passes were bound by
                                         // exactly that staging, 9 600 blocks of it per pass
constexpr int kBoxD = 18;                // 3x3 matrix, upper planes, lower planes, centre x y, bounding radius
constexpr int kLdsBoxes = 32;            // scene boxes of a query kept in LDS by the sample chain
constexpr double kCos1 = 0x1.ffec097f5af8ap-1;   // np.cos(np.deg2rad(1)), find_spot.py:52-59
constexpr double kSin1 = 0x1.1df0b2b89dd1ep-6;   // np.sin(np.deg2rad(1))
constexpr float kDegPerRad = 57.29577951308232f;
constexpr int kMapWindowSide = 256;                       // cells per side of the map patch kept in LDS
constexpr int kMapWindowWords = kMapWindowSide * kMapWindowSide / 32;

struct PlaceWs {
  double *cx, *cy;              // [Q][360] box centre after step r
  double *quat;                 // [Q][360][4] box orientation after step r
  double *rotm;                 // [Q][360][9] its matrix as cut_bounding_box builds it
  unsigned long long *dmin;     // [Q][360] min squared distance of a surface point to the centre
  int32_t *kstar;               // [Q][360] index of the first search radius that holds surface, -1 none
  double *road;                 // [Q][360] mean surface height
  double *planes;               // [Q][360][6] the sample box's upper / lower planes per axis
  int32_t *surf_n;              // [Q][360] number of surface points inside the step's radius
  double *surf_sum, *surf_abs;  // [Q][360] sum of their heights and of the magnitudes, in arrival order
  int32_t *surf_lsb;            // [Q][360] smallest exponent of a last mantissa bit among the heights
  int32_t *surf_list_n;         // [Q][360] entries appended to the ordered list (stops growing at kCap)
  unsigned long long *surf;     // [Q][360][kCap] (label rank << 40 | point index)
  uint32_t *hit;                // [Q][12] bit r: a non-surface scene point is inside the box of step r
  unsigned long long *gather_sq;// [Q] largest squared radius any step of the query needs (bits of a double)
  int32_t *bad;                 // [Q] 1: the descriptor cannot be followed (R3D_PS_BAD_DESCRIPTOR): every kernel leaves the query alone
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
  w.surf_sum = c.take<double>(qr);
  w.surf_abs = c.take<double>(qr);
  w.surf_lsb = c.take<int32_t>(qr);
  w.surf_list_n = c.take<int32_t>(qr);
  w.surf = c.take<unsigned long long>(qr * kCap);
  w.hit = c.take<uint32_t>((size_t)nq * 12);
  w.gather_sq = c.take<unsigned long long>((size_t)nq);
  w.bad = c.take<int32_t>((size_t)nq);
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
    in = in & (lhs < up[a]) & (lhs > dn[a]);
  }
  return in;
}

// Diagnostic builds (-DR3D_GUARD, tools/build_flavour.sh guard -DR3D_GUARD): a kernel that is about to follow the pointers of
// a query descriptor looks at them first; one that cannot be a device address is printed with the head of the descriptor
// and the workgroup leaves (a descriptor overwritten by a stray store shows up as a line of text, not as a queue abort).
#ifdef R3D_GUARD
__device__ __forceinline__ bool odd_pointer(const void *p, bool nullable = false, unsigned long long align = 4) {
  const unsigned long long a = (unsigned long long)p;
  if (!a) return !nullable;
  return a < (1ull << 20) || a >= (1ull << 48) || (a & (align - 1ull));
}
__device__ __noinline__ bool odd_query(const r3d_place_query_t *Q, int q, int kernel, bool say) {
  const r3d_place_query_t &d = Q[q];
  const bool bad = odd_pointer(d.scene) || odd_pointer(d.orig) || odd_pointer(d.boxes, true) || odd_pointer(d.sample) ||
                   odd_pointer(d.map, false, 1) || odd_pointer(d.scene_ranges, true) || odd_pointer(d.orig_ranges, true) ||
                   d.scene_ld != 4 || d.orig_ld != 4 || d.n_scene < 0 || d.n_scene > (1 << 24) || d.n_orig < 0 ||
                   d.n_orig > (1 << 24) || d.m <= 0 || d.m > 8192 || d.n_boxes < 0 || d.n_boxes > 4096;
  if (bad && say) {
    const unsigned long long *u = reinterpret_cast<const unsigned long long *>(&d);
    printf("R3D_GUARD kernel %d query %d at %p: %016llx %016llx %016llx %016llx %016llx %016llx %016llx | n %lld %lld ld %d %d m %d boxes %d | "
           "tail %016llx %016llx %016llx\n", kernel, q, (const void *)&d, u[0], u[1], u[2], u[3], u[4], u[5], u[6], (long long)d.n_scene,
           (long long)d.n_orig, d.scene_ld, d.orig_ld, d.m, d.n_boxes, u[56], u[60], u[62]);
  }
  return bad;
}
#define R3D_GUARD_QUERY(kernel, say) if (odd_query(Q, q, kernel, say)) return
#else
#define R3D_GUARD_QUERY(kernel, say)
#endif

// What k_place_centres looks at before any kernel follows a descriptor's pointers: a descriptor that was never filled, was
// filled for another layout of the struct or has been overwritten shows up as R3D_PS_BAD_DESCRIPTOR in the query's status
// (no placements), not as a fault of the queue.
__device__ __forceinline__ bool address_ok(const void *p, unsigned long long align, bool may_be_null) {
  const unsigned long long a = (unsigned long long)p;
  if (!a) return may_be_null;
  return a >= 4096ull && a < (1ull << 48) && !(a & (align - 1ull));
}
__device__ __forceinline__ bool descriptor_ok(const r3d_place_query_t &d) {
  bool ok = d.n_scene >= 0 && d.n_orig >= 0 && d.n_scene <= (1ll << 40) && d.n_orig <= (1ll << 40);
  const bool slab = d.flavour & R3D_PQ_SCENE_SLAB, orig_slab = d.flavour & R3D_PQ_ORIG_SLAB;
  ok = ok && (slab || (d.scene_ld >= 3 && d.scene_ld <= 4096 && d.scene_label_col >= 0 && d.scene_label_col < d.scene_ld));
  ok = ok && (orig_slab || (d.orig_ld >= 3 && d.orig_ld <= 4096 && d.orig_label_col >= 0 && d.orig_label_col < d.orig_ld));
  ok = ok && d.n_boxes >= 0 && d.m >= 1 && d.m <= kCB * 16 && d.map_rows >= 0 && d.map_cols >= 0;
  ok = ok && d.n_ok_labels >= 0 && d.n_ok_labels <= R3D_PLACE_MAX_OK_LABELS && d.cand_cap >= 0 && d.cand_off >= 0 && d.cand_stride >= 0;
  ok = ok && address_ok(d.scene, slab ? 16 : 8, d.n_scene == 0) && address_ok(d.orig, orig_slab ? 16 : 8, d.n_orig == 0);
  if (orig_slab) ok = ok && address_ok(d.orig_label, 4, d.n_orig == 0);
  ok = ok && address_ok(d.boxes, 8, d.n_boxes == 0) && address_ok(d.sample, 8, false);
  ok = ok && address_ok(d.map, 1, d.map_rows == 0 || d.map_cols == 0);
  ok = ok && address_ok(d.scene_ranges, 4, true) && address_ok(d.orig_ranges, 4, true);
  if (slab) {
    ok = ok && d.scene_head >= 0 && d.scene_head <= d.n_scene;
    ok = ok && address_ok(d.scene_label, 4, d.n_scene == 0) && address_ok(d.scene_alive, 8, d.n_scene == 0);
    ok = ok && address_ok(d.scene_tail_ref, 4, d.scene_head == d.n_scene) && address_ok(d.scene_log5, 8, d.scene_head == d.n_scene);
  }
  return ok;
}

// ---- k_place_centres / k_place_orient: find_spot.py:52-70 applied 360 times to the annotation -----
// Two chains that do not depend on each other: the box centre (three fused multiply-adds per step: 5 us for the 360
// steps) is what the point passes need -- which points can be near which step --, the orientation (matrix, product,
// back to a quaternion, normalised: 0.8 us per step, 0.30 ms of pure latency per call) only what the planes of the
// step's box and the candidates' annotations need.  The orientation runs on a stream of its own beside the point passes.
__global__ void k_place_centres(const r3d_place_query_t *Q, int nq, PlaceWs w, int32_t *status) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const r3d_place_query_t &qq = Q[q];
  double c0 = qq.anno[0], c1 = qq.anno[1], c2 = qq.anno[2];
  bool finite = true;
  for (int i = 0; i < 10; ++i) finite = finite && isfinite(qq.anno[i]);
  for (int i = 0; i < 8; ++i) finite = finite && isfinite(qq.pose[i]);
  const bool followable = descriptor_ok(qq);
  w.bad[q] = followable ? 0 : 1;
  status[q] = (finite ? 0 : R3D_PS_NONFINITE) | (followable ? 0 : R3D_PS_BAD_DESCRIPTOR);
  w.gather_sq[q] = 0ull;
  const double Z[9] = {kCos1, -kSin1, 0.0, kSin1, kCos1, 0.0, 0.0, 0.0, 1.0};
  // (eight steps' centres leave together, 64 bytes of a lane's own row at a time: a store per step touched 64 lines per wave
  // instruction, two instructions per step -- 19 of the kernel's 29 us)
  static_assert(kRot % 8 == 0, "the centres are written eight steps at a time");
  for (int r0 = 0; r0 < kRot; r0 += 8) {
    double x8[8], y8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      double n0 = fma(Z[2], c2, fma(Z[0], c0, Z[1] * c1));      // np.dot(z_rot_matrix, position), :66-70
      double n1 = fma(Z[5], c2, fma(Z[3], c0, Z[4] * c1));
      double n2 = fma(Z[8], c2, fma(Z[6], c0, Z[7] * c1));
      c0 = n0;
      c1 = n1;
      c2 = n2;
      x8[u] = c0;
      y8[u] = c1;
    }
    const size_t o = (size_t)q * kRot + r0;                     // (kRot * 8 bytes per query: rows are 64-byte aligned for r0 % 8 == 0)
    double2 *px = reinterpret_cast<double2 *>(w.cx + o), *py = reinterpret_cast<double2 *>(w.cy + o);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      px[u] = make_double2(x8[2 * u], x8[2 * u + 1]);
      py[u] = make_double2(y8[2 * u], y8[2 * u + 1]);
    }
  }
}

__global__ void k_place_orient(const r3d_place_query_t *Q, int nq, PlaceWs w) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq || w.bad[q]) return;
  const r3d_place_query_t &qq = Q[q];
  Quat a{qq.anno[3], qq.anno[4], qq.anno[5], qq.anno[6]};
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
    size_t o = (size_t)q * kRot + r;
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
  if (w.bad[q]) return;
  R3D_GUARD_QUERY(__LINE__, j == 0);
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
  o[15] = b[0];
  o[16] = b[1];
  o[17] = sqrt(b[7] * b[7] / 4 + b[8] * b[8] / 4 + b[9] * b[9]) * 1.001 + 0.01;   // no box point is further from the centre
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

// A cloud row: x y z at columns 0-2 and the label at `label_col` of rows `ld` doubles apart.  Rows
// packed as [x y z label] (ld == 4) are read with two 16-byte loads.
struct Pt {
  double x, y, z, label;
};
// Pointers that come out of a descriptor are generic to the compiler; every array a descriptor
// points to lives in global memory, and saying so gives global_load instead of flat_load (which
// also waits on the LDS counter).
template <class T>
using InGlobal = const __attribute__((address_space(1))) T *;
template <class T>
__device__ __forceinline__ InGlobal<T> in_global(const T *p) {
  return (InGlobal<T>)p;
}

__device__ __forceinline__ Pt load_point(const double *base, int64_t i, int ld, int label_col) {
  Pt p;
  if (ld == 4 && label_col == 3) {
    typedef double pair_t __attribute__((ext_vector_type(2)));
    InGlobal<pair_t> v = (InGlobal<pair_t>)(base + i * 4);
    pair_t a = v[0], b = v[1];
    p.x = a.x;
    p.y = a.y;
    p.z = b.x;
    p.label = b.y;
  } else {
    InGlobal<double> r = in_global(base + i * ld);
    p.x = r[0];
    p.y = r[1];
    p.z = r[2];
    p.label = r[label_col];
  }
  return p;
}

// A point of the ORIGINAL cloud: float64 rows, or (R3D_PQ_ORIG_SLAB) float32 rows + label words -- the same values.
__device__ __forceinline__ Pt load_orig(const r3d_place_query_t &qq, int64_t i) {
  if (qq.flavour & R3D_PQ_ORIG_SLAB) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const f4_t f = ((InGlobal<f4_t>)reinterpret_cast<const f4_t *>(qq.orig))[i];
    Pt p;
    p.x = (double)f.x;
    p.y = (double)f.y;
    p.z = (double)f.z;
    p.label = (double)(in_global(qq.orig_label)[i] & 0xFFFFu);
    return p;
  }
  return load_point(qq.orig, i, qq.orig_ld, qq.orig_label_col);
}
__device__ __forceinline__ double load_orig_z(const r3d_place_query_t &qq, int64_t i) {
  if (qq.flavour & R3D_PQ_ORIG_SLAB) return (double)in_global(reinterpret_cast<const float *>(qq.orig))[i * 4 + 2];
  return in_global(qq.orig)[i * qq.orig_ld + 2];
}

// Distance range (from the sensor axis) of every 64-point chunk of a cloud: a scan in its native
// order (ring by ring) has narrow ranges, and a point pass can skip the chunks that cannot be
// within reach of the circle the sample moves on.  Rounded outwards; [0, inf) if anything is odd.
template <class T>
__global__ __launch_bounds__(kPB) void k_chunk_ranges(const T *rows, int64_t n, int ld, float *ranges) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * (kPB / 64) + (threadIdx.x >> 6);
  if (chunk * 64 >= n) return;
  const int64_t i = chunk * 64 + lane;
  float lo = INFINITY, hi = 0.f;
  bool odd = false;
  if (i < n) {
    double x = (double)rows[i * ld], y = (double)rows[i * ld + 1];
    float rho = sqrtf((float)(x * x + y * y));
    odd = !isfinite(rho);
    lo = hi = rho;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  odd = __any(odd);
  if (lane == 0) {
    ranges[chunk * 2 + 0] = odd ? 0.f : lo * (1.f - 1e-6f);
    ranges[chunk * 2 + 1] = odd ? INFINITY : hi * (1.f + 1e-6f);
  }
}

// True when no point of the chunk can be within `reach` of the circle of radius rho_c.
__device__ __forceinline__ bool chunk_out_of_reach(const float *ranges, int64_t chunk, float rho_c, float reach) {
  if (!ranges) return false;
  float lo = in_global(ranges)[chunk * 2], hi = in_global(ranges)[chunk * 2 + 1];
  return lo > rho_c + reach || hi < rho_c - reach;
}

// The chunks of a block's 4096 points that are within reach, listed in LDS by the first wave with one
// coalesced load of the ranges (a serial skip test per chunk would pay one memory latency each).
static_assert(kPointsPerBlock <= 64 * 64 && kPointsPerBlock % 64 == 0, "one ballot covers the chunks of a block");
__device__ __forceinline__ int list_chunks_in_reach(const float *ranges, int64_t start, int64_t end, float rho_c,
                                                    float reach, unsigned char *s_chunk, int *s_count) {
  if (threadIdx.x < 64) {
    int64_t c0 = start + (int64_t)threadIdx.x * 64;
    bool take = c0 < end && !chunk_out_of_reach(ranges, c0 >> 6, rho_c, reach);
    unsigned long long m = __ballot(take);
    if (take) s_chunk[__popcll(m & ((1ull << threadIdx.x) - 1ull))] = (unsigned char)threadIdx.x;
    if (threadIdx.x == 0) *s_count = __popcll(m);
  }
  __syncthreads();
  return *s_count;
}

// The same before a block has staged anything (one turn per block): the radius of the circle from the first step's centre
// as the kernels compute it from their tables.  -1: the block takes several turns and lists per turn.
__device__ __forceinline__ int chunks_before_tables(const float *ranges, int64_t start0, int64_t n, double cx0, double cy0,
                                                    float reach, unsigned char *s_chunk, int *s_count) {
  if (kTurns != 1) return -1;
  const float rho_c = sqrtf((float)(cx0 * cx0 + cy0 * cy0));
  const int64_t end = start0 + kPointsPerBlock < n ? start0 + kPointsPerBlock : n;
  return list_chunks_in_reach(ranges, start0, end, rho_c, reach, s_chunk, s_count);
}

__device__ __forceinline__ int label_rank(const r3d_place_query_t &qq, double label) {
  for (int j = 0; j < qq.n_ok_labels; ++j)
    if (label == (double)qq.ok_labels[j]) return j;
  return -1;
}

// f(r) for the steps r of the circular window [first, first + count) whose bit is set in a 360-bit mask in LDS: a point of
// the wide passes has some 60 steps in reach, most of them already settled -- the set bits are walked, not the window.
template <class F>
__device__ __forceinline__ void for_marked_steps(const uint32_t *mask, int first, int count, F f) {
  auto span = [&](int a, int b) {                                  // steps [a, b), 0 <= a < b <= kRot
    const int w0 = a >> 5, w1 = (b - 1) >> 5;
    for (int wd = w0; wd <= w1; ++wd) {
      uint32_t m = mask[wd];
      if (wd == w0) m &= ~0u << (a & 31);
      if (wd == w1 && (b & 31)) m &= ~(~0u << (b & 31));
      while (m) {
        const int bit = __ffs((int)m) - 1;
        m &= m - 1;
        f((wd << 5) + bit);
      }
    }
  };
  if (count <= 0) return;
  if (count > kRot) count = kRot;
  const int end = first + count;
  if (end <= kRot) {
    span(first, end);
  } else {
    span(first, kRot);
    if (end - kRot > 0) span(0, end - kRot);
  }
}

// The query descriptor is read many times per point (labels, strides): stage it in LDS.
__device__ __forceinline__ void stage_query(r3d_place_query_t &dst, const r3d_place_query_t *src) {
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d = reinterpret_cast<uint32_t *>(&dst);
  for (int i = threadIdx.x; i < (int)(sizeof(r3d_place_query_t) / 4); i += blockDim.x) d[i] = s[i];
  __syncthreads();
}

// ---- k_place_road_min: correct_height's distance test (find_spot.py:123) for every step at once ---
// mode 0: search within `reach` for every step.  mode 1: only the steps whose minimum is still above
// resolved_sq (the narrow pass found nothing that close, so it may have missed the true minimum).
__global__ __launch_bounds__(kPB) void k_place_road_min(const r3d_place_query_t *Q, PlaceWs w, float reach, int mode,
                                                        double resolved_sq) {
  const int q = blockIdx.x, tid = threadIdx.x;        // queries of one scene are neighbours: they share the chunk in L2
  if (w.bad[q]) return;
  R3D_GUARD_QUERY(__LINE__, threadIdx.x == 0 && blockIdx.y == 0);
  const int64_t n = Q[q].n_orig, start0 = (int64_t)blockIdx.y * kPointsPerBlock * kTurns;
  if (start0 >= n) return;
  __shared__ r3d_place_query_t qq;
  __shared__ unsigned char s_chunk[64];
  __shared__ int s_nchunk;
  __shared__ double s_cx[kRot], s_cy[kRot];
  __shared__ unsigned long long s_min[kRot];
  __shared__ unsigned char s_need[kRot];
  // most blocks hold no chunk within reach of the circle the centre runs along (a scan in ring order: a few beams cross
  // it): they leave before the steps' tables are staged
  const int early = chunks_before_tables(Q[q].orig_ranges, start0, n, w.cx[(size_t)q * kRot], w.cy[(size_t)q * kRot], reach, s_chunk,
                                         &s_nchunk);
  if (early == 0) return;
  __shared__ uint32_t s_open[(kRot + 31) / 32];
  if (tid < (kRot + 31) / 32) s_open[tid] = 0u;
  __syncthreads();
  int any = 0;
  for (int r = tid; r < kRot; r += kPB) {
    size_t o = (size_t)q * kRot + r;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
    s_min[r] = R3D_SENT;
    int need = mode == 0 ? 1 : (w.dmin[o] > depth_key(resolved_sq) ? 1 : 0);
    s_need[r] = (unsigned char)need;
    if (need) atomicOr(&s_open[r >> 5], 1u << (r & 31));
    any |= need;
  }
  if (!__syncthreads_or(any)) return;
  stage_query(qq, Q + q);
  const float rho_c = sqrtf((float)(s_cx[0] * s_cx[0] + s_cy[0] * s_cy[0])), th1 = atan2f((float)s_cy[0], (float)s_cx[0]);
  for (int turn = 0; turn < kTurns; ++turn) {
    const int64_t start = start0 + (int64_t)turn * kPointsPerBlock;
    if (start >= n) break;
    const int64_t end = start + kPointsPerBlock < n ? start + kPointsPerBlock : n;
    __syncthreads();                                              // (the previous turn's chunk list is no longer read)
    const int n_chunks = early >= 0 ? early : list_chunks_in_reach(qq.orig_ranges, start, end, rho_c, reach, s_chunk, &s_nchunk);
    for (int a = tid >> 6; a < n_chunks; a += kPB / 64) {         // one 64-point chunk per wave and turn
      const int64_t i = start + (int64_t)s_chunk[a] * 64 + (tid & 63);
      if (i >= end) continue;
      const Pt p = load_orig(qq, i);
      const double x = p.x, y = p.y;
      if (!(p.z > -3.0)) continue;                                // :133-134
      if (label_rank(qq, p.label) < 0) continue;                  // :125-131
      int first, count;
      if (!steps_in_reach((float)x, (float)y, rho_c, th1, reach, first, count)) continue;
      auto look = [&](int r) {
        double dx = x - s_cx[r], dy = y - s_cy[r];
        unsigned long long key = depth_key(dx * dx + dy * dy);   // :123, non-negative: bits are ordered
        if (key < s_min[r]) atomicMin(&s_min[r], key);
      };
      // (the wide pass: some 60 steps in a point's reach, most of them settled by the narrow one -- their bits are walked,
      // 293 -> 197 us per 320 queries; with every step open the counted loop is the faster one: 71 against 156 us)
      if (mode == 0) {
        for (int t = 0; t < count; ++t) {
          int r = first + t;
          look(r >= kRot ? r - kRot : r);
        }
      } else {
        for_marked_steps(s_open, first, count, look);
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < kRot; r += kPB)
    if (s_min[r] != R3D_SENT) atomicMin(&w.dmin[(size_t)q * kRot + r], s_min[r]);
}

// First radius of the growing search that holds surface (find_spot.py:121-140).
__global__ void k_place_kstar(int nq, PlaceWs w, Radii rad) {
  const int t_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = t_raw < nq * kRot;                        // (nobody leaves: the wave reductions below want every lane)
  const int t = live ? t_raw : nq * kRot - 1;
  unsigned long long key = w.dmin[t];
  int k = -1;
  if (live && key != R3D_SENT) {
    double d2 = key_depth(key);
    for (int j = 0; j < rad.n; ++j)
      if (d2 <= rad.sq[j]) {
        k = j;
        break;
      }
  }
  if (live) {
    w.kstar[t] = k;
    w.surf_n[t] = 0;
    w.surf_sum[t] = 0.0;
    w.surf_abs[t] = 0.0;
    w.surf_lsb[t] = INT32_MAX;
    w.surf_list_n[t] = 0;
  }
  // (the largest radius any step of the query needs.  A wave of 64 steps belongs to at most two queries, lane 0's and lane
  // 63's: one maximum each, one atomic each -- 360 atomics on one address per query took their turns in L2: 42 us for 320
  // queries)
  const unsigned long long mine = k >= 0 ? depth_key(rad.sq[k]) : 0ull;
  const int q = t / kRot;
  const int q_first = __shfl(q, 0, 64), q_last = __shfl(q, 63, 64);
  const unsigned long long m_first = wave_max_u64(q == q_first ? mine : 0ull);
  const unsigned long long m_last = wave_max_u64(q == q_last ? mine : 0ull);
  if ((threadIdx.x & 63) == 0) {
    if (m_first) atomicMax(&w.gather_sq[q_first], m_first);
    if (q_last != q_first && m_last) atomicMax(&w.gather_sq[q_last], m_last);
  }
}

// Exponent of the last set mantissa bit of a finite non-zero double: v is a multiple of 2^that.
__device__ __forceinline__ int last_bit_exponent(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  int e = (int)((u >> 52) & 0x7FF);
  unsigned long long mant = u & ((1ull << 52) - 1);
  if (e == 0) return -1074 + (mant ? __ffsll((long long)mant) - 1 : 0);     // subnormal
  mant |= 1ull << 52;
  return e - 1075 + (__ffsll((long long)mant) - 1);
}

// ---- k_place_surface_gather: the points of `surface` (find_spot.py:123-134) of every step ---------
// np.mean adds the heights row by row.  If all of them are multiples of 2^L and the sum of their
// magnitudes stays below 2^(L+53), every partial sum in ANY order is exactly representable, so
// the sum does not depend on the order and atomics may collect it (always so for LiDAR heights,
// which are float32 values of similar size).  The ordered list is kept for the other case.
__global__ __launch_bounds__(kPB) void k_place_surface_gather(const r3d_place_query_t *Q, PlaceWs w, Radii rad) {
  const int q = blockIdx.x, tid = threadIdx.x;        // queries of one scene are neighbours: they share the chunk in L2
  if (w.bad[q]) return;
  R3D_GUARD_QUERY(__LINE__, threadIdx.x == 0 && blockIdx.y == 0);
  const int64_t n = Q[q].n_orig, start0 = (int64_t)blockIdx.y * kPointsPerBlock * kTurns;
  if (start0 >= n) return;
  const double reach_sq = key_depth(w.gather_sq[q]);
  if (!(reach_sq > 0.0)) return;                                  // no step found surface
  __shared__ r3d_place_query_t qq;
  __shared__ unsigned char s_chunk[64];
  __shared__ int s_nchunk;
  __shared__ double s_cx[kRot], s_cy[kRot], s_thr[kRot], s_sum[kRot], s_abs[kRot];
  __shared__ int s_cnt[kRot], s_lsb[kRot];
  const int early = chunks_before_tables(Q[q].orig_ranges, start0, n, w.cx[(size_t)q * kRot], w.cy[(size_t)q * kRot],
                                         (float)sqrt(reach_sq) * 1.01f + 0.05f, s_chunk, &s_nchunk);
  if (early == 0) return;
  for (int r = tid; r < kRot; r += kPB) {
    size_t o = (size_t)q * kRot + r;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
    int k = w.kstar[o];
    s_thr[r] = k >= 0 ? rad.sq[k] : -1.0;
    s_sum[r] = s_abs[r] = 0.0;
    s_cnt[r] = 0;
    s_lsb[r] = INT32_MAX;
  }
  stage_query(qq, Q + q);
  const float reach = (float)sqrt(reach_sq) * 1.01f + 0.05f;
  const float rho_c = sqrtf((float)(s_cx[0] * s_cx[0] + s_cy[0] * s_cy[0])), th1 = atan2f((float)s_cy[0], (float)s_cx[0]);
  for (int turn = 0; turn < kTurns; ++turn) {
  const int64_t start = start0 + (int64_t)turn * kPointsPerBlock;
  if (start >= n) break;
  const int64_t end = start + kPointsPerBlock < n ? start + kPointsPerBlock : n;
  __syncthreads();                                                // (the previous turn's chunk list is no longer read)
  const int n_chunks = early >= 0 ? early : list_chunks_in_reach(qq.orig_ranges, start, end, rho_c, reach, s_chunk, &s_nchunk);
  for (int a = tid >> 6; a < n_chunks; a += kPB / 64) {
    const int64_t i = start + (int64_t)s_chunk[a] * 64 + (tid & 63);
    if (i >= end) continue;
    const Pt p = load_orig(qq, i);
    const double x = p.x, y = p.y;
    if (!(p.z > -3.0)) continue;
    int rank = label_rank(qq, p.label);
    if (rank < 0) continue;
    int first, count;
    if (!steps_in_reach((float)x, (float)y, rho_c, th1, reach, first, count)) continue;
    for (int t = 0; t < count; ++t) {
      int r = first + t;
      r = r >= kRot ? r - kRot : r;
      double dx = x - s_cx[r], dy = y - s_cy[r];
      if (!(dx * dx + dy * dy <= s_thr[r])) continue;             // :123 with the step's radius
      atomicAdd(&s_cnt[r], 1);
      atomicAdd(&s_sum[r], p.z);
      atomicAdd(&s_abs[r], fabs(p.z));
      if (p.z != 0.0) atomicMin(&s_lsb[r], last_bit_exponent(p.z));
      // the ordered list only matters while it is complete: stop appending once it is full
      size_t o = (size_t)q * kRot + r;
      if (__hip_atomic_load(&w.surf_list_n[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kCap) {
        int slot = atomicAdd(&w.surf_list_n[o], 1);
        if (slot < kCap) w.surf[o * kCap + slot] = ((unsigned long long)rank << 40) | (unsigned long long)i;
      }
    }
  }
  }
  __syncthreads();
  for (int r = tid; r < kRot; r += kPB) {
    if (!s_cnt[r]) continue;
    size_t o = (size_t)q * kRot + r;
    atomicAdd(&w.surf_n[o], s_cnt[r]);
    atomicAdd(&w.surf_sum[o], s_sum[r]);
    atomicAdd(&w.surf_abs[o], s_abs[r]);
    atomicMin(&w.surf_lsb[o], s_lsb[r]);
  }
}

// ---- k_place_road_level: np.mean(surface, axis=0)[2] (find_spot.py:144) and the box planes --------
// One wave per step.  The surface points are put in the reference's order (label order of the
// config, then point order) by ranking their keys against each other, and their heights are
// added one by one, like np.mean over the rows of `surface` does.
__global__ __launch_bounds__(kPB) void k_place_road_level(const r3d_place_query_t *Q, int nq, PlaceWs w,
                                                          int32_t *status) {
  __shared__ double s_z[kPB / 64][kCap];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t = blockIdx.x * (kPB / 64) + wv;
  const bool live = t < nq * kRot;
  const int q = live ? t / kRot : 0;
  const r3d_place_query_t &qq = Q[q];
  int n = 0, n_all = 0;
  bool exact = false;
  double sum_any_order = 0.0;
  if (live && w.kstar[t] >= 0) {
    n = n_all = w.surf_n[t];
    sum_any_order = w.surf_sum[t];
    const double mag = w.surf_abs[t];
    const int lsb = w.surf_lsb[t];
    // mag < 2^(ilogb(mag)+1); exact when that exponent is at most lsb + 53
    exact = lsb == INT32_MAX || (isfinite(mag) && ilogb(mag) + 1 - lsb <= 53);
    if (exact) n = 0;                                              // no need to order anything
    if (n > kCap) {
      if (lane == 0) atomicOr(&status[q], R3D_PS_SURFACE_OVERFLOW);
      n = kCap;
    }
  }
  const unsigned long long *l = w.surf + (size_t)(live ? t : 0) * kCap;
  static_assert(kCap == 128, "two keys per lane");
  const unsigned long long k0 = lane < n ? l[lane] : R3D_SENT, k1 = lane + 64 < n ? l[lane + 64] : R3D_SENT;
  const unsigned long long idx_mask = (1ull << 40) - 1;
  const double z0 = lane < n ? load_orig_z(qq, (int64_t)(k0 & idx_mask)) : 0.0;
  const double z1 = lane + 64 < n ? load_orig_z(qq, (int64_t)(k1 & idx_mask)) : 0.0;
  int r0 = 0, r1 = 0;
  for (int j = 0; j < n; ++j) {                                   // keys are distinct (point indices are)
    unsigned long long kj = j < 64 ? __shfl(k0, j, 64) : __shfl(k1, j - 64, 64);
    r0 += kj < k0 ? 1 : 0;
    r1 += kj < k1 ? 1 : 0;
  }
  if (lane < n) s_z[wv][r0] = z0;
  if (lane + 64 < n) s_z[wv][r1] = z1;
  __syncthreads();
  if (!live || lane != 0) return;
  double road = 0.0;
  if (exact && n_all > 0) {
    road = sum_any_order / (double)n_all;
  } else if (n > 0) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += s_z[wv][i];
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
  const int q = blockIdx.x, tid = threadIdx.x;
  if (w.bad[q]) return;
  R3D_GUARD_QUERY(__LINE__, threadIdx.x == 0 && blockIdx.y == 0);
  const int64_t n = Q[q].n_scene, start0 = (int64_t)blockIdx.y * kPointsPerBlock * kTurns;
  if (start0 >= n) return;
  __shared__ r3d_place_query_t qq;
  __shared__ unsigned char s_chunk[64];
  __shared__ int s_nchunk;
  __shared__ double s_cx[kRot], s_cy[kRot];
  __shared__ float s_boxf[kRot][12];                // float copy of the step's box: matrix (9), centre (3)
  __shared__ unsigned char s_near[kRot];
  __shared__ uint32_t s_hit[12];
  int early;
  {
    const double l0 = Q[q].anno[7], w0 = Q[q].anno[8], h0 = Q[q].anno[9];
    early = chunks_before_tables(Q[q].scene_ranges, start0, n, w.cx[(size_t)q * kRot], w.cy[(size_t)q * kRot],
                                 (float)sqrt(l0 * l0 / 4 + w0 * w0 / 4 + h0 * h0) * 1.01f + 0.05f, s_chunk, &s_nchunk);
  }
  if (early == 0) return;
  int any = 0;
  for (int r = tid; r < kRot; r += kPB) {
    size_t o = (size_t)q * kRot + r;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
    int near = w.kstar[o] >= 0;
    s_near[r] = (unsigned char)near;
    any |= near;
    for (int i = 0; i < 9; ++i) s_boxf[r][i] = (float)w.rotm[o * 9 + i];
    s_boxf[r][9] = (float)w.cx[o];
    s_boxf[r][10] = (float)w.cy[o];
    s_boxf[r][11] = (float)w.road[o];
  }
  if (tid < 12) s_hit[tid] = 0u;
  if (!__syncthreads_or(any)) return;
  stage_query(qq, Q + q);
  const double l = qq.anno[7], wd = qq.anno[8], h = qq.anno[9];
  const double reach_d = sqrt(l * l / 4 + wd * wd / 4 + h * h);     // no box point is further from the centre
  const float reach = (float)reach_d * 1.01f + 0.05f;
  const float hl = (float)l * 0.5f, hw = (float)wd * 0.5f, hh = (float)h;
  const bool od_label = qq.flavour & R3D_PQ_COLLIDE_LABEL, od_above = qq.flavour & R3D_PQ_COLLIDE_ABOVE;
  const float rho_c = sqrtf((float)(s_cx[0] * s_cx[0] + s_cy[0] * s_cy[0])), th1 = atan2f((float)s_cy[0], (float)s_cx[0]);
  // float32 rounding of coordinates up to ~100 m is ~1e-5 m; the planes sit at |column|^2 * size, within
  // 1e-12 of size for the unit quaternions of the chain
  const float slack = 1e-3f + 1e-5f * (rho_c + reach + (float)fabs(qq.anno[2]) + 10.f);
  for (int turn = 0; turn < kTurns; ++turn) {
  const int64_t start = start0 + (int64_t)turn * kPointsPerBlock;
  if (start >= n) break;
  const int64_t end = start + kPointsPerBlock < n ? start + kPointsPerBlock : n;
  __syncthreads();                                                // (the previous turn's chunk list is no longer read)
  const int n_chunks = early >= 0 ? early : list_chunks_in_reach(qq.scene_ranges, start, end, rho_c, reach, s_chunk, &s_nchunk);
  for (int a = tid >> 6; a < n_chunks; a += kPB / 64) {
    const int64_t i = start + (int64_t)s_chunk[a] * 64 + (tid & 63);
    if (i >= end) continue;
    Pt p;
    if (qq.flavour & R3D_PQ_SCENE_SLAB) {
      // the scene as it stands in its batch (include/real3daug_hip.h): dead points skipped, float32-exact head points from
      // the slab, inserted points' float64 coordinates from the log -- what r3d_batch_export_rows would have written
      if (!((in_global(qq.scene_alive)[i >> 6] >> (i & 63)) & 1ull)) continue;
      if (i < qq.scene_head) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t f = ((InGlobal<f4_t>)reinterpret_cast<const f4_t *>(qq.scene))[i];
        p.x = (double)f.x, p.y = (double)f.y, p.z = (double)f.z;
      } else {
        InGlobal<double> row = in_global(qq.scene_log5 + (int64_t)in_global(qq.scene_tail_ref)[i - qq.scene_head] * 5);
        p.x = row[0], p.y = row[1], p.z = row[2];
      }
      p.label = (double)(in_global(qq.scene_label)[i] & 0xFFFFu);
    } else {
      p = load_point(qq.scene, i, qq.scene_ld, qq.scene_label_col);
    }
    const double x = p.x, y = p.y, z = p.z;
    if (od_label ? !(p.label == (double)qq.collide_label)         // OD :120-121
                 : label_rank(qq, p.label) >= 0) continue;        // SS :94-95: surface may be inside the box
    int first, count;
    if (!steps_in_reach((float)x, (float)y, rho_c, th1, reach, first, count)) continue;
    const float xf = (float)x, yf = (float)y, zf = (float)z;
    for (int t = 0; t < count; ++t) {
      int r = first + t;
      r = r >= kRot ? r - kRot : r;
      if (!s_near[r]) continue;
      // float32 look at the point in the box's frame, with slack: only what may be inside goes to the
      // float64 planes of the reference
      const float *bf = s_boxf[r];
      float dx = xf - bf[9], dy = yf - bf[10], dz = zf - bf[11];
      float u0 = bf[0] * dx + bf[3] * dy + bf[6] * dz, u1 = bf[1] * dx + bf[4] * dy + bf[7] * dz,
            u2 = bf[2] * dx + bf[5] * dy + bf[8] * dz;
      if (!(fabsf(u0) < hl + slack && fabsf(u1) < hw + slack && u2 > -slack && u2 < hh + slack)) continue;
      size_t o = (size_t)q * kRot + r;
      if (od_above && !(z >= w.road[o] + qq.collide_dz)) continue;  // OD :123-124, the box bottom is the road level
      if (inside_box(w.rotm + o * 9, w.planes + o * 6, w.planes + o * 6 + 3, x, y, z))
        atomicOr(&s_hit[r >> 5], 1u << (r & 31));
    }
  }
  }
  __syncthreads();
  if (tid < 12 && s_hit[tid]) atomicOr(&w.hit[(size_t)q * 12 + tid], s_hit[tid]);
}

// Points per thread with which the sample chain takes a sample of m points.
// (as few as the workgroup's 512 threads allow: a step of the chain is the instruction stream of ONE wave -- its points'
// float64 chains, some 100 instructions per point -- and a workgroup's waves run side by side on the CU's four SIMDs; with 4
// points per lane a pedestrian was two waves of 550 instructions per step, 2.0 us)
__host__ __device__ inline int chain_class(int m) {
  int need = (m + kCB - 1) / kCB;
  // measured per class on 320 queries (us per launch, 4 points per lane -> this choice): pedestrians of 400 points 717 -> 591
  // with one point per lane (7 waves), cars of 1 500 points 1 046 -> 991 with three (8 waves); cyclists of 600 points are
  // better off with four per lane on three waves (819) than with two on five (881): more waves, longer barriers
  return need <= 1 ? 1 : need == 3 ? 3 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
}
// threads of the sample chain's workgroup that take part: whole waves, PPT points per lane
template <int PPT>
__device__ __forceinline__ int act_threads(int m) {
  int t = (((m + PPT - 1) / PPT) + 63) & ~63;
  return t < 64 ? 64 : (t > kCB ? kCB : t);
}

// ---- k_place_sample_chain: the loop of find_spot.py:228-269 on the sample's points ---------------
// One workgroup per query; a thread keeps PPT points in registers through the 360 steps.  Per step:
// rotate, look the map cell up (a patch of the map sits in LDS as one bit per cell), agree on
// "all on allowed surface" across the workgroup, apply the height correction, test the points
// against the scene boxes that can touch the sample at this step.  What a step needs from the
// earlier kernels (road level, near flag, scene-in-box bit, box centre) is in LDS.  The per-point
// code is branch-free over all PPT slots (slots past the sample's end compute on zeros and are
// masked out of the votes), so the float64 chains of different points overlap.
struct ChainLds {
  uint32_t allowed[kMapWindowWords];   // bit = the map cell is an allowed surface, window of the map
  double road[kRot], cx[kRot], cy[kRot];
  double box[kLdsBoxes * kBoxD];
  unsigned char near[kRot], flags[kRot], vote[3][kRot];
  unsigned short rot[kRot];
  uint32_t hit[12];
  float red[2][kCB / 64];
};

// The chain's per-step agreement travels through LDS only: wait for the LDS counter and meet.  (__syncthreads() also waits
// for every global store in flight -- the candidate clouds a possible step has just written, a microsecond or two each
// time, 96 times per query on the bench's frames.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int PPT, bool POINTWISE, bool ONE = false>
__device__ __forceinline__ void sample_chain(ChainLds &lds, const r3d_place_query_t *Q, const PlaceWs &w, int max_boxes,
                                             uint8_t *flags, int32_t *n_possible, int32_t *rot_out,
                                             double *anno_out, double *cand, int32_t first_cand, int32_t *status) {
  const int q = blockIdx.x, tid = threadIdx.x;
  R3D_GUARD_QUERY(__LINE__, threadIdx.x == 0);
  const r3d_place_query_t &qq = Q[q];
  const int m = qq.m;
  uint32_t *s_allowed = lds.allowed;
  double *s_road = lds.road, *s_cx = lds.cx, *s_cy = lds.cy, *s_box = lds.box;
  unsigned char *s_near = lds.near, *s_flags = lds.flags;
  unsigned char(*s_vote)[kRot] = lds.vote;
  unsigned short *s_rot = lds.rot;
  uint32_t *s_hit = lds.hit;
  float(*s_red)[kCB / 64] = lds.red;
  double x[PPT], y[PPT], z[PPT], c3[PPT], c4[PPT];   // (c3, c4: intensity and label, written with every candidate)
  bool valid[PPT];
  int bad_input = 0;
  float rho2 = 0.f, ext2 = 0.f;                     // largest distance^2 from the sensor / from the box centre
  const double ax = qq.anno[0], ay = qq.anno[1];
  // The 360 steps are a chain (a step's height correction moves every point for the later ones), and every step has the
  // workgroup agree once or twice: its cost is the barrier among the waves, not the arithmetic.  So only as many waves
  // stay as the sample needs at PPT points per lane -- one wave for a pedestrian of 250 points: no barrier left at all --
  // and the others leave once the tables are staged (a barrier waits for the surviving waves only).
  const int T = act_threads<PPT>(m);
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    int i = tid + u * T;
    x[u] = y[u] = z[u] = c3[u] = c4[u] = 0.0;
    valid[u] = i < m && tid < T;
    if (valid[u]) {
      x[u] = in_global(qq.sample)[(size_t)i * 5 + 0];
      y[u] = in_global(qq.sample)[(size_t)i * 5 + 1];
      z[u] = in_global(qq.sample)[(size_t)i * 5 + 2];
      c3[u] = in_global(qq.sample)[(size_t)i * 5 + 3];
      c4[u] = in_global(qq.sample)[(size_t)i * 5 + 4];
      if (!(isfinite(x[u]) && isfinite(y[u]) && isfinite(z[u]))) bad_input = 1;
      rho2 = fmaxf(rho2, (float)(x[u] * x[u] + y[u] * y[u]));
      ext2 = fmaxf(ext2, (float)((x[u] - ax) * (x[u] - ax) + (y[u] - ay) * (y[u] - ay)));
    }
  }
  if (bad_input) atomicOr(&status[q], R3D_PS_NONFINITE);
  for (int r = tid; r < kRot; r += kCB) {
    size_t o = (size_t)q * kRot + r;
    s_road[r] = w.road[o];
    s_near[r] = w.kstar[o] >= 0 ? 1 : 0;
    s_vote[0][r] = s_vote[1][r] = s_vote[2][r] = 0;
    s_cx[r] = w.cx[o];
    s_cy[r] = w.cy[o];
  }
  if (tid < 12) s_hit[tid] = w.hit[(size_t)q * 12 + tid];
  const int nb = qq.n_boxes;
  const double *gboxes = w.boxes + (size_t)q * max_boxes * kBoxD;
  const bool lds_boxes = nb <= kLdsBoxes;
  if (lds_boxes)
    for (int i = tid; i < nb * kBoxD; i += kCB) s_box[i] = gboxes[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    rho2 = fmaxf(rho2, __shfl_xor(rho2, o, 64));
    ext2 = fmaxf(ext2, __shfl_xor(ext2, o, 64));
  }
  if ((tid & 63) == 0) {
    s_red[0][tid >> 6] = rho2;
    s_red[1][tid >> 6] = ext2;
  }
  __syncthreads();
  const double T00 = qq.pose[0], T01 = qq.pose[1], T02 = qq.pose[2], T03 = qq.pose[3];
  const double T10 = qq.pose[4], T11 = qq.pose[5], T12 = qq.pose[6], T13 = qq.pose[7];
  const double mv0 = qq.map_move[0], mv1 = qq.map_move[1];
  const int rows = qq.map_rows, cols = qq.map_cols;
  InGlobal<uint8_t> map = in_global(qq.map);
  const uint64_t okm0 = qq.ok_map[0], okm1 = qq.ok_map[1], okm2 = qq.ok_map[2], okm3 = qq.ok_map[3];
  auto allowed = [&](unsigned v) -> bool {
    uint64_t wv = v < 64 ? okm0 : v < 128 ? okm1 : v < 192 ? okm2 : okm3;
    return (wv >> (v & 63)) & 1ull;
  };
  const bool od_map = qq.flavour & R3D_PQ_MAP_NEEDS_POINT;
  const double map_lo = od_map ? 0.0 : -1.0;         // OD: 0 <= position; SS: int() of (-1, 0) is cell 0
  const int cand_cap = qq.cand_cap;
  const int64_t cand_stride = qq.cand_stride;
  double *cand_q = cand + qq.cand_off;
  float rho = 0.f, ext = 0.f;
  for (int i = 0; i < kCB / 64; ++i) {
    rho = fmaxf(rho, s_red[0][i]);
    ext = fmaxf(ext, s_red[1][i]);
  }
  rho = sqrtf(rho);
  ext = sqrtf(ext) * 1.001f + 0.01f;                 // no sample point is further from the box centre (in x, y)
  // The sample stays on a circle around the sensor, so it only ever looks at a patch of the map:
  // that patch goes to LDS as one bit per cell (cells outside it are still read from memory).
  int wr0, wc0, wh, ww;
  {
    double r0 = rho * sqrt(T00 * T00 + T01 * T01) + fabs(T02) * 8.0 + 2.0;
    double r1 = rho * sqrt(T10 * T10 + T11 * T11) + fabs(T12) * 8.0 + 2.0;
    double lo0 = floor(T03 - mv0 - r0), lo1 = floor(T13 - mv1 - r1);
    lo0 = lo0 < 0.0 ? 0.0 : lo0;
    lo1 = lo1 < 0.0 ? 0.0 : lo1;
    double hi0 = T03 - mv0 + r0 + 1.0, hi1 = T13 - mv1 + r1 + 1.0;
    hi0 = hi0 > (double)rows ? (double)rows : hi0;
    hi1 = hi1 > (double)cols ? (double)cols : hi1;
    bool none = !(hi0 > lo0 && hi1 > lo1);                        // also for NaN
    wr0 = none ? 0 : (int)lo0;
    wc0 = none ? 0 : (int)lo1;
    int h = none ? 0 : (int)hi0 - wr0, wd = none ? 0 : (int)hi1 - wc0;
    wh = h > kMapWindowSide ? kMapWindowSide : h;
    ww = wd > kMapWindowSide ? kMapWindowSide : wd;
  }
  for (int wi = tid; wi < (wh * ww + 31) / 32; wi += kCB) {
    uint32_t bits = 0u;
    for (int b = 0; b < 32; ++b) {
      int c = wi * 32 + b;
      if (c >= wh * ww) break;
      unsigned v = map[(size_t)(wr0 + c / ww) * cols + wc0 + c % ww];
      bits |= (uint32_t)allowed(v) << b;
    }
    s_allowed[wi] = bits;
  }
  __syncthreads();
  // Whole waves leave here (T is a multiple of 64: act_threads), the others go on meeting at s_barrier: the hardware's
  // barrier counts the waves of the workgroup that have not ended (CDNA ISA, s_barrier: "waves that have terminated are not
  // waited for"), which the HIP programming model does not promise -- hence the barriers below are spelled as the
  // instruction (lds_barrier), not as __syncthreads(), and the two facts they rest on are checked where they are made.
  static_assert(kCB % 64 == 0, "the sample chain's workgroup is whole waves");
  if ((T & 63) != 0 || (long long)PPT * T < m) {              // (cannot happen: chain_class / act_threads)
    if (tid == 0) atomicOr(&status[q], R3D_PS_NONFINITE);
    return;
  }
  if (tid >= T) return;
  double anno_z = qq.anno[2];
  int n_out = 0;
  // (od_map as arithmetic on compare results, the rotation's flavour as a template parameter, the rare map cell outside the
  // LDS patch behind ONE branch after the PPT points: a branch inside the per-point code keeps the compiler from
  // interleaving the points' float64 chains -- each then runs at its full latency, 2.0 us per step for a pedestrian)
  const int od_i = od_map ? 1 : 0;
  for (int r = 0; r < kRot; ++r) {
    int bad = 0, in_any_map = 0, slow = 0;
    double g0[PPT], g1[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      // SS :72  bbox_pcl[:, :3] = (z_rot_matrix @ bbox_pcl[:, :3].T).T   (matrix product: a0*b0, fma, fma)
      // OD :94-99  np.dot(z_rot_matrix, column) per point             (matrix x vector: fma(a2,b2, fma(a0,b0, a1*b1)))
      double nx, ny, nz;
      if (POINTWISE) {
        nx = fma(0.0, z[u], fma(kCos1, x[u], -kSin1 * y[u]));
        ny = fma(0.0, z[u], fma(kSin1, x[u], kCos1 * y[u]));
        nz = fma(1.0, z[u], fma(0.0, x[u], 0.0 * y[u]));
      } else {
        nx = fma(0.0, z[u], fma(-kSin1, y[u], kCos1 * x[u]));
        ny = fma(0.0, z[u], fma(kCos1, y[u], kSin1 * x[u]));
        nz = fma(1.0, z[u], fma(0.0, y[u], 0.0 * x[u]));
      }
      x[u] = nx;
      y[u] = ny;
      z[u] = nz;
      // transformation_matrix @ [x y z 1], minus map_move, astype(int): :234-238
      if (ONE) {
        // a sample of ONE point: both products have a single column and numpy hands them to the matrix x vector
        // routine -- :72 then accumulates as OD's per-point product does (POINTWISE), and the 4 x 4 one as two partial
        // sums of rounded products (even and odd terms) added at the end (oracle/find_spot_oracle.py: blas_matvec4)
        g0[u] = __dadd_rn(__dadd_rn(__dmul_rn(T00, nx), __dmul_rn(T02, nz)), __dadd_rn(__dmul_rn(T01, ny), T03)) - mv0;
        g1[u] = __dadd_rn(__dadd_rn(__dmul_rn(T10, nx), __dmul_rn(T12, nz)), __dadd_rn(__dmul_rn(T11, ny), T13)) - mv1;
      } else {
        g0[u] = fma(T03, 1.0, fma(T02, nz, fma(T01, ny, T00 * nx))) - mv0;
        g1[u] = fma(T13, 1.0, fma(T12, nz, fma(T11, ny, T10 * nx))) - mv1;
      }
    }
    int cell[PPT], i0s[PPT], i1s[PPT];
    int inmap[PPT], inwin[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      // int() truncates: the index is in [0, rows) exactly when -1 < g0 < rows (:240-243); OD: 0 <= position
      const int lo0 = (int)(g0[u] > map_lo) | (od_i & (int)(g0[u] == map_lo));
      const int lo1 = (int)(g1[u] > map_lo) | (od_i & (int)(g1[u] == map_lo));
      inmap[u] = lo0 & (int)(g0[u] < (double)rows) & lo1 & (int)(g1[u] < (double)cols);
      i0s[u] = inmap[u] ? (int)g0[u] : 0;
      i1s[u] = inmap[u] ? (int)g1[u] : 0;
      const int a = i0s[u] - wr0, b = i1s[u] - wc0;
      inwin[u] = (int)(a >= 0) & (int)(a < wh) & (int)(b >= 0) & (int)(b < ww);
      cell[u] = inwin[u] ? a * ww + b : 0;
    }
    uint32_t wd[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) wd[u] = s_allowed[cell[u] >> 5];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int v = valid[u] ? 1 : 0;
      const int ok = (int)((wd[u] >> (cell[u] & 31)) & 1u);
      in_any_map |= v & inmap[u];
      bad |= v & inmap[u] & inwin[u] & (ok ^ 1);                  // :245-248
      slow |= v & inmap[u] & (inwin[u] ^ 1);
    }
    if (slow) {                                                   // a cell outside the patch in LDS: rare
#pragma unroll
      for (int u = 0; u < PPT; ++u)
        if (valid[u] && inmap[u] && !inwin[u] && !allowed(map[(size_t)i0s[u] * cols + i1s[u]])) bad = 1;
    }
    if (bad) s_vote[0][r] = 1;                                    // one slot per step: no reset, one barrier
    if (od_map && in_any_map) s_vote[2][r] = 1;
    lds_barrier();
    const bool on_surface = !s_vote[0][r] && (!od_map || s_vote[2][r]);
    const bool near = s_near[r];
    if (on_surface && near) {                                     // correct_height, :142-148
      double road = s_road[r];
      double z_move = road - anno_z;
#pragma unroll
      for (int u = 0; u < PPT; ++u) z[u] += z_move;
      anno_z = road;
    }
    const bool scene_hit = (s_hit[r >> 5] >> (r & 31)) & 1u;
    bool sample_hit = false;
    if (on_surface && near && !scene_hit && nb > 0) {             // :99-103
      // only boxes whose bounding circle reaches the sample's can hold one of its points
      const float cxr = (float)s_cx[r], cyr = (float)s_cy[r];
      int in_any = 0, tested = 0;
      auto test_boxes = [&](const double *boxes) {                // called with an LDS or a global pointer:
        for (int b = 0; b < nb; ++b) {                            // keeps the address space known at both sites
          const double *bx = boxes + (size_t)b * kBoxD;
          float dx = (float)bx[15] - cxr, dy = (float)bx[16] - cyr, reach = (float)bx[17] + ext;
          if (dx * dx + dy * dy > reach * reach) continue;        // uniform across the workgroup
          tested = 1;
#pragma unroll
          for (int u = 0; u < PPT; ++u) in_any |= (valid[u] && inside_box(bx, bx + 9, bx + 12, x[u], y[u], z[u])) ? 1 : 0;
        }
      };
      if (lds_boxes) test_boxes(s_box);
      else test_boxes(gboxes);
      if (tested) {
        if (in_any) s_vote[1][r] = 1;
        lds_barrier();
        sample_hit = s_vote[1][r];
      }
    }
    const bool possible = on_surface && near && !scene_hit && !sample_hit;
    if (tid == 0)
      s_flags[r] = (unsigned char)((on_surface ? R3D_PF_ON_SURFACE : 0) | (near ? R3D_PF_NEAR_ROAD : 0) |
                                   (scene_hit ? R3D_PF_SCENE_IN_BOX : 0) | (sample_hit ? R3D_PF_SAMPLE_IN_BOX : 0) |
                                   (possible ? R3D_PF_POSSIBLE : 0));
    if (possible) {                                               // :257-264
      int j = n_out - first_cand;
      if (j >= 0 && j < cand_cap) {
        double *out = cand_q + (size_t)j * cand_stride;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
          int i = tid + u * T;
          if (!valid[u]) continue;
          out[(size_t)i * 5 + 0] = x[u];
          out[(size_t)i * 5 + 1] = y[u];
          out[(size_t)i * 5 + 2] = z[u];
          out[(size_t)i * 5 + 3] = c3[u];                        // (from registers: a load here waits for every store
          out[(size_t)i * 5 + 4] = c4[u];                        // of the steps before -- loads and stores share a counter)
        }
      }
      if (tid == 0) s_rot[n_out] = (unsigned short)r;
      ++n_out;
    }
  }
  lds_barrier();                                                  // (s_flags / s_rot of thread 0; surviving waves only)
  // the annotation of a possible placement: centre (cx, cy, road level of that step), orientation
  for (int r = tid; r < kRot; r += T) flags[(size_t)q * kRot + r] = s_flags[r];
  for (int j = tid; j < n_out; j += T) {
    int r = s_rot[j];
    size_t o = (size_t)q * kRot + r, oo = (size_t)q * kRot + j;
    rot_out[oo] = r + 1;
    anno_out[oo * 7 + 0] = s_cx[r];
    anno_out[oo * 7 + 1] = s_cy[r];
    anno_out[oo * 7 + 2] = s_road[r];
    for (int i = 0; i < 4; ++i) anno_out[oo * 7 + 3 + i] = w.quat[o * 4 + i];
  }
  if (tid == 0) n_possible[q] = n_out;
}

// Samples of up to 2048 points (every class of the reference's object database) in one launch; the
// workgroup picks the instance of its size class.  Larger samples: the second kernel.
__global__ __launch_bounds__(kCB) void k_place_sample_chain(const r3d_place_query_t *Q, PlaceWs w, int max_boxes,
                                                           uint8_t *flags, int32_t *n_possible, int32_t *rot_out,
                                                           double *anno_out, double *cand, int32_t first_cand,
                                                           int32_t *status) {
  __shared__ ChainLds lds;
  if (w.bad[blockIdx.x]) {                                        // (R3D_PS_BAD_DESCRIPTOR: no placements)
    if (threadIdx.x == 0) n_possible[blockIdx.x] = 0;
    return;
  }
  const bool pointwise = Q[blockIdx.x].flavour & R3D_PQ_POINTWISE_ROTATION;
  if (Q[blockIdx.x].m == 1) {                                     // one column: the matrix x vector arithmetic throughout
    sample_chain<1, true, true>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status);
    return;
  }
  switch (chain_class(Q[blockIdx.x].m)) {
#define R3D_CHAIN_CASE(P)                                                                                                  \
  case P:                                                                                                                   \
    if (pointwise) sample_chain<P, true>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status); \
    else sample_chain<P, false>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status);      \
    break;
    R3D_CHAIN_CASE(1)
    R3D_CHAIN_CASE(3)
    R3D_CHAIN_CASE(4)
#undef R3D_CHAIN_CASE
    default: break;
  }
}

__global__ __launch_bounds__(kCB) void k_place_sample_chain_large(const r3d_place_query_t *Q, PlaceWs w, int max_boxes,
                                                                 uint8_t *flags, int32_t *n_possible,
                                                                 int32_t *rot_out, double *anno_out, double *cand,
                                                                 int32_t first_cand, int32_t *status) {
  __shared__ ChainLds lds;
  if (w.bad[blockIdx.x]) {                                        // (R3D_PS_BAD_DESCRIPTOR: no placements)
    if (threadIdx.x == 0) n_possible[blockIdx.x] = 0;
    return;
  }
  const bool pointwise = Q[blockIdx.x].flavour & R3D_PQ_POINTWISE_ROTATION;
  switch (chain_class(Q[blockIdx.x].m)) {
    case 8:
      if (pointwise) sample_chain<8, true>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status);
      else sample_chain<8, false>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status);
      break;
    case 16:
      if (pointwise) sample_chain<16, true>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status);
      else sample_chain<16, false>(lds, Q, w, max_boxes, flags, n_possible, rot_out, anno_out, cand, first_cand, status);
      break;
    default: break;
  }
}

}  // namespace

// A stream of its own per (device, caller's stream) for k_place_orient, with the two events that tie it to the caller's:
// made on first use, kept until r3d_places_release() (calls on one stream follow each other in stream order, so the events
// can be recorded again and again).  nullptr: the runtime would not make one (the chain then runs in line).
struct SideStream {
  hipStream_t stream;
  hipEvent_t start, done;
};
static std::mutex g_side_mu;
static std::map<std::pair<int, hipStream_t>, SideStream *> g_side;
static SideStream *side_stream_of(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_side_mu);
  auto it = g_side.find({dev, st});
  if (it != g_side.end()) return it->second;
  SideStream *s = new SideStream();
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->start, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess) {
    delete s;
    s = nullptr;
  }
  g_side[{dev, st}] = s;
  return s;
}

extern "C" int r3d_places_release(void) {
  std::lock_guard<std::mutex> lock(g_side_mu);
  int now = 0;
  const bool have_dev = hipGetDevice(&now) == hipSuccess;
  for (auto &kv : g_side) {
    SideStream *s = kv.second;
    if (!s) continue;
    if (have_dev) (void)hipSetDevice(kv.first.first);
    (void)hipStreamSynchronize(s->stream);
    (void)hipEventDestroy(s->start);
    (void)hipEventDestroy(s->done);
    (void)hipStreamDestroy(s->stream);
    delete s;
  }
  g_side.clear();
  if (have_dev) (void)hipSetDevice(now);
  return R3D_OK;
}

extern "C" size_t r3d_places_workspace_bytes(int32_t n_queries, int32_t max_boxes) {
  if (n_queries <= 0 || max_boxes < 0) return 0;
  return carve_places(n_queries, max_boxes, nullptr).total;
}

extern "C" int r3d_places_chunk_ranges(const double *rows, int64_t n, int32_t ld, float *ranges, void *stream) {
  if (!rows || !ranges || n < 0 || ld < 2) return fail(R3D_E_ARG, "places_chunk_ranges: bad argument");
  if (n == 0) return R3D_OK;
  int64_t chunks = (n + 63) / 64;
  hipLaunchKernelGGL(k_chunk_ranges<double>, dim3((unsigned)((chunks + kPB / 64 - 1) / (kPB / 64))), dim3(kPB), 0,
                     static_cast<hipStream_t>(stream), rows, n, (int)ld, ranges);
  R3D_LAUNCHED("k_chunk_ranges");
  return R3D_OK;
}

extern "C" int r3d_places_chunk_ranges_f32(const float *rows4, int64_t n, float *ranges, void *stream) {
  if (!rows4 || !ranges || n < 0) return fail(R3D_E_ARG, "places_chunk_ranges_f32: bad argument");
  if (n == 0) return R3D_OK;
  int64_t chunks = (n + 63) / 64;
  hipLaunchKernelGGL(k_chunk_ranges<float>, dim3((unsigned)((chunks + kPB / 64 - 1) / (kPB / 64))), dim3(kPB), 0,
                     static_cast<hipStream_t>(stream), rows4, n, 4, ranges);
  R3D_LAUNCHED("k_chunk_ranges");
  return R3D_OK;
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
  if (max_m > kCB * 16) return fail(R3D_E_ARG, "places: samples are limited to 8192 points");
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
  hipLaunchKernelGGL(k_place_centres, dim3((n_queries + 63) / 64), dim3(64), 0, st, queries, n_queries, w, status);
  // the orientation chain beside the point passes, on the helper stream that belongs to `st` (no helper stream: in line)
  // (calls that share one stream must come from one host thread at a time: the two events of the stream's helper are
  // recorded again by every call)
  SideStream *side = side_stream_of(st);
  // whatever way this function is left once the side launch is out, the caller's stream waits for it: the workspace and the
  // queries must not be freed or reused while k_place_orient still writes w.quat
  struct JoinSide {
    SideStream *side;
    hipStream_t st;
    bool armed = false, joined = false;
    void join() {
      if (armed && !joined) (void)hipStreamWaitEvent(st, side->done, 0);
      joined = true;
    }
    ~JoinSide() { join(); }
  } join_side{side, st};
  if (side) {
    R3D_HIP(hipEventRecord(side->start, st));
    R3D_HIP(hipStreamWaitEvent(side->stream, side->start, 0));
    hipLaunchKernelGGL(k_place_orient, dim3((n_queries + 63) / 64), dim3(64), 0, side->stream, queries, n_queries, w);
    if (hipEventRecord(side->done, side->stream) != hipSuccess) {
      (void)hipStreamSynchronize(side->stream);                  // (no event to wait for: wait here)
      return fail(R3D_E_HIP, "places: hipEventRecord on the helper stream");
    }
    join_side.armed = true;
  } else {
    hipLaunchKernelGGL(k_place_orient, dim3((n_queries + 63) / 64), dim3(64), 0, st, queries, n_queries, w);
  }
  if (max_boxes > 0)
    hipLaunchKernelGGL(k_place_boxes, dim3((n_queries * max_boxes + 255) / 256), dim3(256), 0, st, queries,
                       n_queries, max_boxes, w);
  const int64_t per_block = (int64_t)kPointsPerBlock * kTurns;
  const int pb_orig = (int)((max_n_orig + per_block - 1) / per_block);
  const int pb_scene = (int)((max_n_scene + per_block - 1) / per_block);
  // distance passes of growing reach (metres); a pass only serves the steps whose minimum the
  // previous one could not settle (nothing found within its reach)
  // (round 5: 0.6 m, then the full reach for the steps still open -- the 1.8 m pass in between cost more than it saved the
  // last one: 1.80 -> 1.75 ms per 320 queries, 4.10 -> 3.94 per 1 280)
  constexpr int two_pass = 1;
  const double reach_of_pass[3] = {0.6, two_pass ? sqrt(reach_max) : 1.8, sqrt(reach_max)};
  if (pb_orig > 0) {
    double settled = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
      double reach_m = reach_of_pass[pass] < sqrt(reach_max) ? reach_of_pass[pass] : sqrt(reach_max);
      if (pass > 0 && reach_m <= settled) break;
      hipLaunchKernelGGL(k_place_road_min, dim3(n_queries, pb_orig), dim3(kPB), 0, st, queries, w,
                         (float)(reach_m * 1.01 + 0.05), pass == 0 ? 0 : 1, settled * settled);
      settled = reach_m;
    }
  }
  hipLaunchKernelGGL(k_place_kstar, dim3((unsigned)((qr + 255) / 256)), dim3(256), 0, st, n_queries, w, rad);
  if (pb_orig > 0)
    hipLaunchKernelGGL(k_place_surface_gather, dim3(n_queries, pb_orig), dim3(kPB), 0, st, queries, w, rad);
  if (side) {                                                      // from here on the steps' orientations are read
    join_side.joined = true;
    R3D_HIP(hipStreamWaitEvent(st, side->done, 0));
  }
  hipLaunchKernelGGL(k_place_road_level, dim3((unsigned)((qr + kPB / 64 - 1) / (kPB / 64))), dim3(kPB), 0, st, queries,
                     n_queries, w, status);
  if (pb_scene > 0)
    hipLaunchKernelGGL(k_place_scene_in_box, dim3(n_queries, pb_scene), dim3(kPB), 0, st, queries, w);
  hipLaunchKernelGGL(k_place_sample_chain, dim3(n_queries), dim3(kCB), 0, st, queries, w, max_boxes, flags,
                     n_possible, rot_out, anno_out, cand, first_cand, status);
  if (chain_class(max_m) > 4)
    hipLaunchKernelGGL(k_place_sample_chain_large, dim3(n_queries), dim3(kCB), 0, st, queries, w, max_boxes, flags,
                       n_possible, rot_out, anno_out, cand, first_cand, status);
  R3D_LAUNCHED("placement kernels");
  return R3D_OK;
}

// =====================================================================================
// cut_bounding_box for many boxes at once (tools/cut_bbox.py:7-68; the object-database
// creation of cut_object/cut_out.py:100-157 calls it once per annotated object of a frame).
// Per box: the indices of the cloud's points inside it, in cloud order.  Pass 1 leaves one
// bit per (box, point) and per-tile counts, pass 2 turns the counts of a box into offsets,
// pass 3 writes the indices at offset + rank.
// =====================================================================================
namespace {
constexpr int kCutTile = 2048;          // points per block: 256 threads x 8

struct CutBox {
  double R[9], up[3], dn[3];
  double label;                         // NaN: any label
  float cx, cy, cz, reach;              // bounding sphere around the (bottom) centre
};

__global__ void k_cut_prepare(const double *boxes10, const double *labels, int k, int strict, CutBox *out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= k) return;
  const double *q = boxes10 + (size_t)b * 10;
  CutBox c;
  quat_to_matrix(quat_normalize(Quat{q[3], q[4], q[5], q[6]}), c.R);
  box_planes(c.R, q[0], q[1], q[2], q[7], q[8], q[9], c.up, c.dn);
  c.label = labels ? labels[b] : NAN;
  c.cx = (float)q[0];
  c.cy = (float)q[1];
  c.cz = (float)q[2];
  c.reach = (float)sqrt(q[7] * q[7] / 4 + q[8] * q[8] / 4 + q[9] * q[9]) * 1.001f + 0.01f;
  (void)strict;
  out[b] = c;
}

// strict: cut_bounding_box keeps lhs < up and lhs > dn (:30-64); separate_bbox (:71-123) calls a
// point outside when lhs > up or lhs < dn, so its box keeps the faces (and NaN rows).
__device__ __forceinline__ bool cut_inside(const CutBox &c, double x, double y, double z, int strict) {
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double lhs = c.R[a] * x + c.R[3 + a] * y + c.R[6 + a] * z;
    in = in & (strict ? (lhs < c.up[a]) & (lhs > c.dn[a]) : !(lhs > c.up[a]) & !(lhs < c.dn[a]));
  }
  return in;
}

__global__ __launch_bounds__(256) void k_cut_flags(const double *rows, int64_t n, int ld, int label_col,
                                                  const CutBox *boxes, int k, int strict,
                                                  unsigned long long *bits, int32_t *tile_cnt, int tiles) {
  __shared__ CutBox s_box;
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * kCutTile;
  const int64_t words = (n + 63) / 64;
  double x[8], y[8], z[8], lab[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    int64_t i = t0 + u * 256 + tid;
    x[u] = y[u] = z[u] = lab[u] = 0.0;
    if (i < n) {
      x[u] = rows[i * ld];
      y[u] = rows[i * ld + 1];
      z[u] = rows[i * ld + 2];
      lab[u] = label_col >= 0 ? rows[i * ld + label_col] : 0.0;
    }
  }
  for (int b = 0; b < k; ++b) {
    __syncthreads();
    if (tid < (int)(sizeof(CutBox) / 4))
      reinterpret_cast<uint32_t *>(&s_box)[tid] = reinterpret_cast<const uint32_t *>(boxes + b)[tid];
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    int mine = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int64_t i = t0 + u * 256 + tid;
      bool in = false;
      if (i < n) {
        float dx = (float)x[u] - s_box.cx, dy = (float)y[u] - s_box.cy, dz = (float)z[u] - s_box.cz;
        // far from the box: outside for sure (a NaN coordinate fails this test and takes the exact one)
        if (!(dx * dx + dy * dy + dz * dz > s_box.reach * s_box.reach)) in = cut_inside(s_box, x[u], y[u], z[u], strict);
        if (in && s_box.label == s_box.label) in = lab[u] == s_box.label;
      }
      unsigned long long m = __ballot(in);
      if (lane == 0 && t0 + u * 256 + wave * 64 < n) {
        bits[(size_t)b * words + ((t0 + u * 256 + wave * 64) >> 6)] = m;
        mine += __popcll(m);
      }
    }
    if (lane == 0) s_cnt[wave] = mine;
    __syncthreads();
    if (tid == 0) tile_cnt[(size_t)b * tiles + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  }
}

__global__ void k_cut_offsets(int32_t *tile_cnt, int tiles, int k, int32_t *counts) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= k) return;
  int run = 0;
  for (int t = 0; t < tiles; ++t) {
    int c = tile_cnt[(size_t)b * tiles + t];
    tile_cnt[(size_t)b * tiles + t] = run;
    run += c;
  }
  counts[b] = run;
}

__global__ __launch_bounds__(256) void k_cut_write(int64_t n, int k, const unsigned long long *bits,
                                                  const int32_t *tile_off, int tiles, int32_t *index,
                                                  int64_t index_cap) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kCutTile, words = (n + 63) / 64;
  if (t0 >= n) return;
  // the tile's 32 words in index order: word w covers points t0 + 64 w ..; every wave takes 8 of them
  int run = tile_off[(size_t)b * tiles + blockIdx.x];
  __shared__ int s_pre[32];
  if (tid < 32) {
    int64_t wd = (t0 >> 6) + tid;
    int c = wd < words ? __popcll(bits[(size_t)b * words + wd]) : 0;
    int inc = c;
    for (int o = 1; o < 32; o <<= 1) {
      int v = __shfl_up(inc, o, 64);
      if (tid >= o) inc += v;
    }
    s_pre[tid] = inc - c;
  }
  __syncthreads();
  for (int w8 = 0; w8 < 8; ++w8) {
    int wi = wave * 8 + w8;
    int64_t wd = (t0 >> 6) + wi;
    if (wd >= words) break;
    unsigned long long m = bits[(size_t)b * words + wd];
    if ((m >> lane) & 1ull) {
      int64_t o = (int64_t)run + s_pre[wi] + __popcll(m & ((1ull << lane) - 1ull));
      if (o < index_cap) index[(size_t)b * index_cap + o] = (int32_t)(wd * 64 + lane);
    }
  }
}
}  // namespace

extern "C" size_t r3d_cut_boxes_workspace_bytes(int64_t n, int32_t k) {
  if (n < 0 || k <= 0) return 0;
  size_t words = (size_t)(n + 63) / 64, tiles = (size_t)(n + kCutTile - 1) / kCutTile;
  return align_up(sizeof(CutBox) * (size_t)k) + align_up(words * 8 * (size_t)k) + align_up((tiles + 1) * 4 * (size_t)k);
}

extern "C" int r3d_cut_boxes(const double *rows, int64_t n, int32_t ld, int32_t label_col, const double *boxes10,
                             const double *box_labels, int32_t k, int32_t strict, int32_t *counts, int32_t *index,
                             int64_t index_cap, void *workspace, size_t workspace_bytes, void *stream) {
  if (!rows || !boxes10 || !counts || !index || !workspace) return fail(R3D_E_ARG, "cut_boxes: null argument");
  if (n <= 0 || n > (int64_t)1 << 31 || ld < 3 || k <= 0 || k > 65535 || label_col >= ld || index_cap <= 0)
    return fail(R3D_E_ARG, "cut_boxes: bad shape");
  if (workspace_bytes < r3d_cut_boxes_workspace_bytes(n, k)) return fail(R3D_E_WORKSPACE, "cut_boxes: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles = (int)((n + kCutTile - 1) / kCutTile);
  Carver c(workspace);
  CutBox *boxes = c.take<CutBox>((size_t)k);
  unsigned long long *bits = c.take<unsigned long long>((size_t)((n + 63) / 64) * k);
  int32_t *tile_cnt = c.take<int32_t>((size_t)(tiles + 1) * k);
  hipLaunchKernelGGL(k_cut_prepare, dim3((k + 63) / 64), dim3(64), 0, st, boxes10, box_labels, (int)k, (int)strict, boxes);
  hipLaunchKernelGGL(k_cut_flags, dim3(tiles), dim3(256), 0, st, rows, n, (int)ld, (int)label_col, boxes, (int)k,
                     (int)strict, bits, tile_cnt, tiles);
  hipLaunchKernelGGL(k_cut_offsets, dim3((k + 63) / 64), dim3(64), 0, st, tile_cnt, tiles, (int)k, counts);
  hipLaunchKernelGGL(k_cut_write, dim3(tiles, k), dim3(256), 0, st, n, (int)k, bits, tile_cnt, tiles, index, index_cap);
  R3D_LAUNCHED("cut_boxes kernels");
  return R3D_OK;
}
