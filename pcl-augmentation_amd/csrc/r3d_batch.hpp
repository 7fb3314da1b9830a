// Shared by the translation units of the batched path (Level 2): r3d_batch.hip (step 0, rebase,
// compaction) and r3d_insert.hip (the insert kernels).
//
// State of a scene between r3d_batch_begin and r3d_batch_finish:
//   pix[i]         pixel id of point i under the scene's current elevation bounds; written by the
//                  projection (step 0 / rebase) and for appended points, never changed otherwise
//   alive bit i    one bit per point, 64 points ("chunk") per word: cleared when the point's pixel
//                  turns visible for an accepted insert (insertion.py:472-473), set for appended points
//   chunk_box[c]   row / column bounding box of the pixels of chunk c (all of its points, dead or alive)
//   tile_alive[t]  living points of the 2048-point tile t: the compaction's offsets without a counting pass
// Round 5, VIRTUAL ORDER: all of the above lean on the point order of LiDAR files (64 consecutive points = a piece of one ring:
// a chunk's box is one row by ~50 columns).  A cloud whose points come in no such order (the reference does not care:
// insertion.py:100-127) gets, at step 0, a permutation of its points by (row, 64-column band) -- n_virt[s] > 0 --, and point
// number j < n_virt[s] of everything above (pix, alive, chunk_box, tile_alive) is point perm[j] of the slabs; the slabs, the
// log and the OUTPUT ORDER are untouched: r3d_batch_finish / export_delta / export_rows first put the alive bits back
// into slab order (k_unvirtual: alive_o, tile_o).
#pragma once

#include "r3d_device.hpp"
#include "r3d_host.hpp"

namespace r3d {

constexpr int kPT = 256;             // threads of the streaming kernels
constexpr int kPerThread = 8;
constexpr int kTile = kPT * kPerThread;   // points per block tile
constexpr int kKeyCap = R3D_MAX_SAMPLE;
constexpr int kMaxChain = 64;        // insert slots of one k_insert_chain launch
constexpr int kRecInts = 16;         // int32 words of a published slot record
constexpr int kParkInts = 32;        // int32 words of the header a parked pair leaves (r3d_insert.hip)
constexpr int kEntry = 32;           // bytes of a chunk-list entry of the insert kernels (r3d_insert.hip)
constexpr int kHoldCap = 16;         // pixels of bound-holding points the image route keeps per scene (r3d_image.hip)

struct BatchWs {
  unsigned long long *qkeys;    // [B][2] ordered keys of min / max of z/r
  int32_t *tile_alive;          // [B*tiles] living points per tile
  uint32_t *cand;               // [B*cand_stride] per-scene scratch lists (slow-projection queue; insert fallbacks)
  int32_t *all_list;            // [B] identity
  int32_t *all_count;           // [1] = B
  unsigned long long *chunk_box; // [B*chunks] rows/cols bounding box of 64 consecutive points
  unsigned long long *alive;    // [B*chunks] alive bit of every point
  double *row_q;                // [B*(rows+2)] c*|c|, c = cos of the row edges (entry k: edge k-1)
  double *col_dir;              // [(cols+1)*2] unit vector of every column edge
  float *row_qf;                // [B*(rows+2)*2] float32 limits of z/r either side of every row edge (k_project's screen)
  float *col_dirf;              // [(cols+1)*2] col_dir in float32
  double *q_ext;                // [B*2] min and max of z/r (the points that hold the elevation bounds)
  int32_t *n_slow;              // [B] points queued for k_project_slow
  int32_t *chain_progress;      // [B] slots of the scene completed by the running k_insert_chain (< 0: see r3d_insert.hip)
  int32_t *n_total0;            // [B] n_total when the running k_insert_chain was launched
  int32_t *defer_from;          // [B] first slot of the launch left to k_insert_big (n_slots: none)
  unsigned long long *alive_shadow; // [B*chunks] the alive bits of the culled copy a REJECTED candidate leaves (min_points < 0)
  int32_t *tile_shadow;         // [B*tiles] ... and its living points per tile
  int32_t *shadow_valid;        // [B] 1: r3d_batch_export_rows shows that copy; r3d_batch_adopt_rejected makes it the scene
  int32_t *super_rows;          // [B*supers*2] first / last row over the boxes of 64 consecutive chunks (clouds of kSuperMinChunks
                                // chunks and more: the chunk list looks at a chunk's box only when its super-box reaches the window)
  int32_t *recs;                // [B*kMaxChain*kRecInts] what every finished slot of the launch publishes
  int32_t *park;                // [B*kMaxChain] hand-over word of every pair of the running launch: its evaluator leaves "parked"
                                // there, the workgroup that finishes the scene's previous slot "predecessor done" -- whoever
                                // comes second carries on with the pair (r3d_insert.hip: nobody ever waits)
  int32_t *park_hdr;            // [B*kMaxChain*kParkInts] what a parked pair leaves for the workgroup that commits it
  int32_t *n_virt;              // [B] > 0: the scene's first n_virt points are in virtual order (see the top of this file)
  int32_t *box_area;            // [B] sum of the areas of the scene's chunk boxes as k_project built them (each capped)
  uint32_t *sort_off;           // [B*sort_blocks*512] per block of 4 096 points and bin: count, then first place (k_virt_hist)
  uint32_t *perm;               // [B*cap] virtual point number -> point of the slabs (first n_virt entries of a scene)
  uint32_t *inv;                // [B*cap] point of the slabs -> virtual point number
  unsigned long long *alive_o;  // [B*chunks] alive words of a scene in virtual order, put back into slab order (k_unvirtual)
  int32_t *tile_o;              // [B*tiles] ... and the living points per tile in slab order
  int32_t *tile_pre;            // [B*tiles] living points in front of every tile (k_tile_prefix: clouds of many tiles)
  unsigned char *glist;         // [B*chunks*kEntry] k_insert_big's chunk lists, one area per scene (a pair of k_insert_chain whose
                                // list exceeds its LDS takes room from the pool)
  unsigned char *tile_pool;     // [pool_bytes] the launch's bump pool: depth tiles / candidate lists / chunk lists / scratch images
                                // of the pairs whose window exceeds a workgroup's LDS
  unsigned long long *pool_head; // [1] bytes handed out in the running launch
  int32_t *queue_next;          // [16] the running k_insert_chain's work queues: next pair of XCD x's queue in [x] (all pairs: [0])
#ifdef R3D_EXP_IMAGE
  // ---- experiment (r3d_image.hip, -DR3D_EXP_IMAGE only): a persistent raw range image per scene ----
  unsigned long long *img;      // [B*npix] minimum SQUARED depth (float64 bits of x*x + y*y + z*z) over the LIVING points of
                                // every pixel (insertion.py:118-125; the root is monotone), R3D_SENT = nobody there
  uint16_t *kstep;              // [B*npix] step of the latest accepted insert that made the pixel visible (0: none): a point
                                // is dead iff its pixel's step exceeds its birth step (insertion.py:470-473 per pixel)
  uint32_t *occ;                // [B*rows*cols/32] occupancy bits of img (the label image of insertion.py:119-120)
  int32_t *img_valid;           // [B] 1: img / kstep / occ / hold_pix describe the scene as it stands
  int32_t *img_dirty;           // [B] 1: kstep holds kills the alive words do not show yet (k_apply_kills)
  int32_t *hold_pix;            // [B*kHoldCap] pixels of the living points that hold an elevation bound (z/r == q_ext)
  int32_t *n_hold;              // [B] how many (more than kHoldCap: any occupied pixel of the first / last row counts)
  uint8_t *band_fall;           // [B*rows] k_image_bands: this row band of the scene listed more chunks than its workgroup holds --
  int32_t *n_fall;              // [B] ... how many of the scene's bands: their points go through k_image_build's global atomics
#endif
  int32_t *dbg;                 // [64] diagnostic counters of the insert kernels (r3d_batch_debug_counters: the first 16;
                                // [16..31]: what a diagnostic build notes about the first failed check, `reset` bit 1 asks for them;
                                // [32..63]: round 5's counters, `reset` bit 2)
  int64_t pool_bytes;
  int64_t cand_stride;          // uint32 entries of `cand` per scene
  size_t total;
};

inline int tiles_of(const r3d_batch_t &b) { return (int)((b.cap + kTile - 1) / kTile); }
inline int chunks_of(const r3d_batch_t &b) { return (int)((b.cap + 63) / 64); }
// Super-boxes: the rows of 64 consecutive chunks (4096 points: a ring or two of a scan in ring order, every column).  Kept
// for clouds of kSuperMinChunks chunks and more (and under diagnostic bit 256): written by k_super_rows after every
// projection of the scene and by rebase_scene, widened by the commit of an insert for the chunks it appends to.
#ifndef R3D_SUPER_MIN_CHUNKS
#define R3D_SUPER_MIN_CHUNKS 4096
#endif
constexpr int kSuperMinChunks = R3D_SUPER_MIN_CHUNKS;
constexpr int kDbgSuper = 256;
constexpr int kCntVirtual = 37;      // BatchWs::dbg: scenes put into virtual order at step 0 (D_VIRTUAL of r3d_insert.hip's counters)
// Round 6, the bin-edge risk counted instead of argued: points whose pixel the reference
// formula decided with the fractional row or column position within kEdgeRisk of an integer -- where NumPy's arctan2 / arccos
// and the device library's, which differ within an ULP, could truncate to neighbouring bins.  [40]: scene points (k_project_slow,
// rebase: everything the verified fast projection could not confirm comes through there), [41]: sample points.
constexpr int kCntEdgeScene = 40, kCntEdgeSample = 41;
// [42]: scenes projected under R3D_B_FILE_ORDER whose chunk boxes say that their points come in NO file order (mean box beyond
// kVirtMeanArea pixels): the promise costs such a scene a walk over the whole cloud per insert, never a result -- counted so
// that a caller (SceneBatch(order="auto") between two looks) can see it.
constexpr int kCntPromiseBroken = 42;
constexpr double kEdgeRisk = 1e-12;
inline int supers_of(const r3d_batch_t &b) { return (chunks_of(b) + 63) / 64; }
#ifdef __HIPCC__
__host__ __device__
#endif
inline bool supers_on(const r3d_batch_t &b, int chunks) { return chunks >= kSuperMinChunks || (b.reserved & kDbgSuper); }

inline BatchWs carve_batch(const r3d_batch_t &b, void *base) {
  BatchWs w;
  Carver c(base);
  int tiles = tiles_of(b);
  w.qkeys = c.take<unsigned long long>((size_t)b.B * 2);
  w.tile_alive = c.take<int32_t>((size_t)b.B * tiles);
  // per-scene scratch: the slow-projection queue (<= cap entries); far pixels and their minima
  int64_t need = 4 * R3D_FAR_CAP + 64;
  w.cand_stride = need > b.cap ? need : b.cap;
  w.cand_stride = (w.cand_stride + 1) & ~(int64_t)1;
  w.cand = c.take<uint32_t>((size_t)b.B * w.cand_stride);
  w.all_list = c.take<int32_t>((size_t)b.B);
  w.all_count = c.take<int32_t>(1);
  w.chunk_box = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.alive = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.row_q = c.take<double>((size_t)b.B * (b.rows + 2));
  w.col_dir = c.take<double>((size_t)(b.cols + 1) * 2);
  w.row_qf = c.take<float>((size_t)b.B * (b.rows + 2) * 2);
  w.col_dirf = c.take<float>((size_t)(b.cols + 1) * 2);
  w.q_ext = c.take<double>((size_t)b.B * 2);
  w.n_slow = c.take<int32_t>((size_t)b.B);
  w.chain_progress = c.take<int32_t>((size_t)b.B);
  w.n_total0 = c.take<int32_t>((size_t)b.B);
  w.defer_from = c.take<int32_t>((size_t)b.B);
  w.super_rows = c.take<int32_t>((size_t)b.B * supers_of(b) * 2);
  w.alive_shadow = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.tile_shadow = c.take<int32_t>((size_t)b.B * tiles);
  w.shadow_valid = c.take<int32_t>((size_t)b.B);
  w.recs = c.take<int32_t>((size_t)b.B * kMaxChain * kRecInts);
  w.park = c.take<int32_t>((size_t)b.B * kMaxChain);
  w.park_hdr = c.take<int32_t>((size_t)b.B * kMaxChain * kParkInts);
  w.n_virt = c.take<int32_t>((size_t)b.B);
  w.box_area = c.take<int32_t>((size_t)b.B);
  w.sort_off = c.take<uint32_t>((size_t)b.B * ((b.cap + 4095) / 4096) * 512);
  w.perm = c.take<uint32_t>((size_t)b.B * b.cap);
  w.inv = c.take<uint32_t>((size_t)b.B * b.cap);
  w.alive_o = c.take<unsigned long long>((size_t)b.B * chunks_of(b));
  w.tile_o = c.take<int32_t>((size_t)b.B * tiles);
  w.tile_pre = c.take<int32_t>((size_t)b.B * tiles);
  w.glist = c.take<unsigned char>((size_t)b.B * chunks_of(b) * kEntry);
  // The launch's pool: depth tiles, candidate lists, chunk lists and scratch images of the pairs whose window exceeds a
  // workgroup's LDS (on the reference's grid a few per cent of the pairs, 10-60 KB each; on a range image several times
  // that size most cars, up to ~1 MB each).  Per scene 1 MB or 16 bytes per point of its slab, between 64 MB and 4 GB per
  // batch; an exhausted pool costs time, not results (tiles in row bands, the pair left to
  // k_insert_big).  Per-lane footprint: INTEGRATION.md.
  const int64_t per_scene = (int64_t)b.cap * 16 > (1 << 20) ? (int64_t)b.cap * 16 : (1 << 20);
  w.pool_bytes = (int64_t)b.B * per_scene;
  w.pool_bytes = w.pool_bytes < (64ll << 20) ? (64ll << 20) : (w.pool_bytes > (4ll << 30) ? (4ll << 30) : w.pool_bytes);
  w.tile_pool = c.take<unsigned char>((size_t)w.pool_bytes);
  w.pool_head = c.take<unsigned long long>(1);
  w.queue_next = c.take<int32_t>(16);
#ifdef R3D_EXP_IMAGE
  {
    const size_t npix = (size_t)b.rows * b.cols;
    w.img = c.take<unsigned long long>((size_t)b.B * npix);
    w.kstep = c.take<uint16_t>((size_t)b.B * npix);
    w.occ = c.take<uint32_t>((size_t)b.B * (npix / 32));
    w.img_valid = c.take<int32_t>((size_t)b.B);
    w.img_dirty = c.take<int32_t>((size_t)b.B);
    w.hold_pix = c.take<int32_t>((size_t)b.B * kHoldCap);
    w.n_hold = c.take<int32_t>((size_t)b.B);
    w.band_fall = c.take<uint8_t>((size_t)b.B * b.rows);
    w.n_fall = c.take<int32_t>((size_t)b.B);
  }
#endif
  w.dbg = c.take<int32_t>(64);
  w.total = c.off;
  return w;
}

int check_batch(const r3d_batch_t *b);
#ifdef R3D_EXP_IMAGE
// r3d_image.hip: round 6's experiment, a persistent raw range image per scene (tools/image_exp/image_build.sh, profiles/r06_image_build.md)
int launch_image_clear(const r3d_batch_t &b, const BatchWs &w, hipStream_t st);
int launch_image_build(const r3d_batch_t &b, const BatchWs &w, hipStream_t st);
int launch_image_bands(const r3d_batch_t &b, const BatchWs &w, hipStream_t st);
#endif

#ifdef __HIPCC__
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
#ifdef R3D_CHECK
    // (diagnostic build: a tail index or log row out of range is flagged in the scene's status, bit 30, and not followed)
    if (i - n_head >= b.log_cap) {
      atomicOr(&b.status[s], 1 << 30);
      i = n_head;
    }
#endif
    int lr = b.tail_ref[(int64_t)s * b.log_cap + (i - n_head)];
#ifdef R3D_CHECK
    if ((unsigned)lr >= (unsigned)b.log_cap) {
      atomicOr(&b.status[s], 1 << 30);
      lr = 0;
    }
#endif
    const double *q = b.log5 + ((int64_t)s * b.log_cap + lr) * 5;
    x = q[0];
    y = q[1];
    z = q[2];
  }
}

// Pixel id of a point as the batched kernels keep it (r3d_batch_t.pix, far_pix): row in the upper, column in the lower
// 16 bits (rows, cols <= 65535: check_batch) -- no division by the run-time column count wherever a pixel is looked up.
__device__ __forceinline__ uint32_t pack_pix(int row, int col) { return ((uint32_t)row << 16) | (uint32_t)col; }
__device__ __forceinline__ int pix_row(uint32_t p) { return (int)(p >> 16); }
__device__ __forceinline__ int pix_col(uint32_t p) { return (int)(p & 0xFFFFu); }

__device__ __forceinline__ unsigned long long pack_box(int rmin, int rmax, int cmin, int cmax) {
  return (unsigned long long)(rmin & 0xFFFF) | ((unsigned long long)(rmax & 0xFFFF) << 16) |
         ((unsigned long long)(cmin & 0xFFFF) << 32) | ((unsigned long long)(cmax & 0xFFFF) << 48);
}

// Row / column bounding box of the pixels of one wave's 64 points.  Four 32-bit minima / maxima: with the
// identity as the DPP `old` operand every step of their reduction is ONE instruction (v_min_u32_dpp ...);
// the packed 16-bit form needs a move per step because VOP3P takes no DPP.
struct BoxAcc {
  unsigned int rlo = 0xFFFFu, rhi = 0u, clo = 0xFFFFu, chi = 0u;
  __device__ __forceinline__ void add(int row, int col) {
    rlo = (unsigned)row < rlo ? (unsigned)row : rlo;
    rhi = (unsigned)row > rhi ? (unsigned)row : rhi;
    clo = (unsigned)col < clo ? (unsigned)col : clo;
    chi = (unsigned)col > chi ? (unsigned)col : chi;
  }
  __device__ __forceinline__ void add_box(unsigned long long bx, int cols) {      // union with a packed box
    int rmin = (int)(bx & 0xFFFF), rmax = (int)((bx >> 16) & 0xFFFF);
    if (rmin > rmax) return;                                            // empty
    int cmin = (int)((bx >> 32) & 0xFFFF), cmax = (int)((bx >> 48) & 0xFFFF);
    if (cmin > cmax) cmin = 0, cmax = cols - 1;                         // an arc across the azimuth seam (wave_pack_arc): all columns
    add(rmin, cmin);
    add(rmax, cmax);
  }
  __device__ __forceinline__ unsigned long long wave_pack() {          // uniform: the box of the wave's 64 lanes
#define R3D_STEP(C, M)                                                                       \
  {                                                                                          \
    unsigned int t0 = (unsigned int)dpp_take<C, M>(-1, (int)rlo), t1 = (unsigned int)dpp_take<C, M>(0, (int)rhi); \
    unsigned int t2 = (unsigned int)dpp_take<C, M>(-1, (int)clo), t3 = (unsigned int)dpp_take<C, M>(0, (int)chi); \
    rlo = t0 < rlo ? t0 : rlo;                                                               \
    rhi = t1 > rhi ? t1 : rhi;                                                               \
    clo = t2 < clo ? t2 : clo;                                                               \
    chi = t3 > chi ? t3 : chi;                                                               \
  }
    R3D_DPP_STEPS(R3D_STEP)
#undef R3D_STEP
    return pack_box(wave_last_i32((int)rlo), wave_last_i32((int)rhi), wave_last_i32((int)clo), wave_last_i32((int)chi));   // rmin > rmax: empty box
  }
  // The same with the columns as the shorter of two arcs of the azimuth circle: [min, max] of the columns themselves, or of
  // the columns turned by half the image -- which is short when the 64 points sit on either side of the seam (columns
  // cols-1 | 0), as the chunk does that holds the end of one ring of a scan and the beginning of the next.  A box whose
  // first column exceeds its last covers [first, cols) and [0, last] (box_touches_cols).  `col`: this lane's column, -1 =
  // none; the row part comes from the accumulator.
  __device__ __forceinline__ unsigned long long wave_pack_arc(int col, int cols) {
    const int half = cols >> 1;
    const unsigned long long plain = wave_pack();
    const int clo_ = (int)((plain >> 32) & 0xFFFF), chi_ = (int)((plain >> 48) & 0xFFFF);
    if (chi_ - clo_ < half) return plain;                    // (an arc across the seam can only be shorter than half the circle)
    int t = col < 0 ? -1 : (col + half >= cols ? col + half - cols : col + half);
    int tlo = wave_min_i32(t < 0 ? 0x7FFFFFFF : t), thi = wave_max_i32(t);
    if (thi < 0 || thi - tlo >= chi_ - clo_) return plain;
    int a = tlo - half, z = thi - half;
    a = a < 0 ? a + cols : a;
    z = z < 0 ? z + cols : z;
    return (plain & 0xFFFFFFFFull) | ((unsigned long long)(a & 0xFFFF) << 32) | ((unsigned long long)(z & 0xFFFF) << 48);
  }
};

// Does a packed box's column part touch [a, b]?  (first column > last column: the arc across the azimuth seam)
__device__ __forceinline__ bool box_touches_cols(int cmin, int cmax, int a, int b) {
  return cmin <= cmax ? (cmin <= b && cmax >= a) : (b >= cmin || a <= cmax);
}

// The reference formula for one point (insertion.py:74-76, :104-116) as a real function call: inlined,
// the float64 atan2 / acos of the device library raise the register count of every kernel that
// contains them by ~70 VGPRs.  ok: bit 0 row in range, bit 1 column in range, bit 2 angles finite,
// bit 3 elevation outside [min_el, max_el], bit 4 the fractional row or column position within kEdgeRisk of an integer.
struct SphBin {
  double r;
  int row, col, ok;
};
static __device__ __attribute__((noinline)) SphBin spherical_bin(double max_el, double min_el, int rows, int cols, double x,
                                                          double y, double z) {
  Binning bn = make_binning(max_el, min_el, rows, cols);
  Sph sp = spherical(x, y, z);
  SphBin o;
  o.r = sp.r;
  o.ok = bin_point(bn, sp.az, sp.el, o.row, o.col);
  if (isfinite(sp.el) && isfinite(sp.az)) o.ok |= 4;
  if (sp.el < min_el || sp.el > max_el) o.ok |= 8;
  {
    const double fr = (sp.el - bn.min_el - 0.00001) / bn.d_el, fc = pymod(sp.az, kTwoPi) / bn.d_az;   // insertion.py:104-105 before int()
    if (fabs(fr - rint(fr)) < kEdgeRisk || fabs(fc - rint(fc)) < kEdgeRisk) o.ok |= 16;
  }
  return o;
}

// One scene point under the reference formula: returns its pixel, accumulates the wave's box, the far
// list and the status flags.
__device__ __forceinline__ int project_point(const r3d_batch_t &b, int s, const Binning &bn, double x,
                                             double y, double z, int &flags, BoxAcc &box, int *n_risk = nullptr) {
  SphBin sb = spherical_bin(bn.max_el, bn.min_el, bn.rows, bn.cols, x, y, z);
  if (n_risk && (sb.ok & 16)) ++*n_risk;
  struct { double r; } sp = {sb.r};
  int row = sb.row, col = sb.col, p = 0;
  int ok = sb.ok;
  if (!(ok & 1)) flags |= (ok & 4) ? R3D_S_ROW_RANGE : R3D_S_NONFINITE;          // assert :110
  else if (!(ok & 2)) flags |= R3D_S_COL_RANGE;                                   // assert :112
  else {
    p = (int)pack_pix(row, col);
    box.add(row, col);
    if (sp.r > R3D_EMPTY_DEPTH) {           // "first hit overwrites the 500": insertion.py:122-125
      int f = atomicAdd(&b.n_far[s], 1);
      if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = p;
      else flags |= R3D_S_FAR_OVERFLOW;
    }
  }
  return p;
}

// Point number j of the insert kernels' numbering -> point of the slabs (n_virt: BatchWs::n_virt[s], 0 for a scene in file order)
__device__ __forceinline__ int orig_of(const BatchWs &w, const r3d_batch_t &b, int s, int n_virt, int j) {
  return j < n_virt ? (int)w.perm[(int64_t)s * b.cap + j] : j;
}

__device__ __forceinline__ bool alive_bit(const BatchWs &w, int chunks, int s, int i) {
  return (w.alive[(int64_t)s * chunks + (i >> 6)] >> (i & 63)) & 1ull;
}

// ---- the verified fast projection's pieces (k_project in r3d_batch.hip explains the margins) -------------------------
__device__ __forceinline__ float guess_acosf(float q) {
  if (fabsf(q) > 0.5f) return acosf(q);                       // steep beams: the library routine
  float q2 = q * q;                                           // asin series, error < 3e-6 for |q| <= 0.5
  float p = fmaf(q2, 0.02237216f, 0.03038194f);
  p = fmaf(p, q2, 0.04464286f);
  p = fmaf(p, q2, 0.075f);
  p = fmaf(p, q2, 0.16666667f);
  return 1.57079637f - fmaf(p * q2, q, q);
}
__device__ __forceinline__ float guess_atan2f(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __builtin_amdgcn_rcpf(mx), t2 = t * t;       // atan on [0, 1], odd polynomial, error ~1e-5 (v_rcp_f32: 1 ulp)
  float p = fmaf(-0.01172120f, t2, 0.05265332f);
  p = fmaf(p, t2, -0.11643287f);
  p = fmaf(p, t2, 0.19354346f);
  p = fmaf(p, t2, -0.33262347f);
  p = fmaf(p, t2, 0.99997726f);
  float a = p * t;
  a = ay > ax ? 1.57079637f - a : a;
  a = x < 0.f ? 3.14159274f - a : a;
  return y < 0.f ? -a : a;
}

// Float64 confirmation of a guessed bin (rg, cg) against the edge tables.
__device__ __forceinline__ bool confirm_bin(const double *__restrict__ row_cc, const double *__restrict__ col_dir,
                                            int rg, int cg, double x, double y, double z, double ss) {
  const double zz = z * fabs(z);
  const double hi = row_cc[rg == 0 ? 0 : rg + 1], lo = row_cc[rg + 2];
  const double ax = col_dir[2 * cg], ay = col_dir[2 * cg + 1], bx = col_dir[2 * cg + 2], by = col_dir[2 * cg + 3];
  const double mr = 4e-12 * ss, mc = 1e-12 * (fabs(x) + fabs(y) + fabs(z));
  // (non-short-circuit on purpose: six compares and five ANDs instead of five branches)
  return (int)(fabs(zz) < 0.999998 * ss) &                    // acos is ill-conditioned at the poles
         (int)(zz < hi * ss - mr) & (int)(zz > lo * ss + mr) & (int)(ax * y - ay * x > mc) &
         (int)(bx * y - by * x < -mc);
}


// ---- rebase: one workgroup re-bases one scene (rare path) ------------------------------------------
// Triggered when an accepted insert may have moved the elevation bounds.  Does, for that scene only,
// what the reference does for every insert (insertion.py:373-375): recompute the bounds over the
// living points and re-project them.  Dead points keep their (stale) pixel id and their cleared
// alive bit; nothing is moved, the loaded input stays intact.  Phases are separated by a
// device-scope fence + barrier because later phases re-read what earlier ones wrote.
__device__ __forceinline__ void phase_sync() {
  __threadfence();
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void rebase_scene(const r3d_batch_t &b, const BatchWs &w, int chunks, const int s,
                                             unsigned long long *s_min, unsigned long long *s_max) {
  const int tid = threadIdx.x;
  // (the caller has fenced; the scalar cache is dropped as well: the counts below may travel through it, and this
  // workgroup read the older ones before its commit)
  asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  const int n = b.n_total[s], n_head = b.n_head[s], n_virt = w.n_virt[s];
  int32_t *pix = b.pix + (int64_t)s * b.cap;
  // (a) bounds (insertion.py:78-79) via the extreme z/r of the living
  unsigned long long lmin = ~0ull, lmax = 0ull;
  int bad = 0;
  for (int i = tid; i < n; i += NT) {
    if (!alive_bit(w, chunks, s, i)) continue;
    double x, y, z;
    load_point(b, s, orig_of(w, b, s, n_virt, i), n_head, x, y, z);
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
    for (int v = 1; v < NT / 64; ++v) {
      lmin = s_min[v] < lmin ? s_min[v] : lmin;
      lmax = s_max[v] > lmax ? s_max[v] : lmax;
    }
    double max_el = acos(ordered_key_inv(lmin)), min_el = acos(ordered_key_inv(lmax));
    b.bounds[2 * s + 0] = max_el;
    b.bounds[2 * s + 1] = min_el;
    w.q_ext[2 * s + 0] = ordered_key_inv(lmin);
    w.q_ext[2 * s + 1] = ordered_key_inv(lmax);
    b.n_far[s] = 0;
    s_min[0] = depth_key(max_el);                       // (for the row table below)
    s_max[0] = depth_key(min_el);
  }
  __syncthreads();
  // the row-edge table of the fast projection under the new bounds (k_prepare's formula): the insert kernels bin their
  // samples against it
  {
    const double max_el = key_depth(s_min[0]), min_el = key_depth(s_max[0]);
    const double d_el = (max_el - min_el) / (double)b.rows;
    for (int k = tid; k < b.rows + 2; k += NT) {
      double edge = min_el + 0.00001 + (double)(k - 1) * d_el;
      edge = edge < 0.0 ? 0.0 : (edge > kPi ? kPi : edge);
      const double c = cos(edge);
      w.row_q[(int64_t)s * (b.rows + 2) + k] = c * fabs(c);
    }
  }
  phase_sync();
  // (b) re-project the living (insertion.py:74-76, :104-116): pixel ids and chunk boxes -- under the bounds this
  // workgroup has just computed (from LDS: not through a cache that may still hold the old ones)
  Binning bn = make_binning(key_depth(s_min[0]), key_depth(s_max[0]), b.rows, b.cols);
  int flags = 0, n_risk = 0;
  for (int i0 = 0; i0 < n; i0 += NT) {
    int i = i0 + tid;
    BoxAcc box;
    if (i < n && alive_bit(w, chunks, s, i)) {
      double x, y, z;
      load_point(b, s, orig_of(w, b, s, n_virt, i), n_head, x, y, z);
      pix[i] = project_point(b, s, bn, x, y, z, flags, box, &n_risk);
    }
    unsigned long long packed = box.wave_pack();
    int c0 = i0 + (tid & ~63);
    if ((tid & 63) == 0 && c0 < n) w.chunk_box[(int64_t)s * chunks + (c0 >> 6)] = packed;
  }
  if (flags) atomicOr(&b.status[s], flags);
  if (n_risk) atomicAdd(&w.dbg[kCntEdgeScene], n_risk);
  phase_sync();
  // ... and the super-boxes over them (a chunk without a living point has left an empty box: rows 0xFFFF .. 0)
  if (supers_on(b, chunks)) {
    const int n_sup = (chunks + 63) >> 6, n_chunks = (n + 63) >> 6;
    for (int sp = tid; sp < n_sup; sp += NT) {
      int r0 = 0x7FFFFFFF, r1 = -1;
      for (int c = sp << 6; c < ((sp + 1) << 6) && c < n_chunks; ++c) {
        const unsigned long long bx = w.chunk_box[(int64_t)s * chunks + c];
        const int a = (int)(bx & 0xFFFF), z = (int)((bx >> 16) & 0xFFFF);
        if (a <= z) {
          r0 = a < r0 ? a : r0;
          r1 = z > r1 ? z : r1;
        }
      }
      w.super_rows[((int64_t)s * n_sup + sp) * 2 + 0] = r0;
      w.super_rows[((int64_t)s * n_sup + sp) * 2 + 1] = r1;
    }
    phase_sync();
  }
}
#endif  // __HIPCC__

}  // namespace r3d
