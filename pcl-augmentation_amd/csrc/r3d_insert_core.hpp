// The (slot, scene) pair of the insert step as both routes of Level 2 use it (r3d_insert.hip: the chain route, r3d_image.hip:
// the image route): the sample phase -- projection of the candidate, its window of the range image, counting sort by (pixel,
// index), min depth per pixel, 5x3 closing --, the chain route's scene phase and commit, the image route's scene phase and
// commit.  Everything lives in one workgroup's LDS; see the top of r3d_insert.hip for the phases.
#pragma once

#include "r3d_batch.hpp"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#ifndef R3D_BIG_WAVES
#define R3D_BIG_WAVES 1
#endif
#ifndef R3D_CHAIN_WAVES
#define R3D_CHAIN_WAVES 4
#endif
// listed points a thread of the gather has in flight at a time.  8 costs the chain kernels 32 - 64 bytes of scratch
// per lane (they sit on their 128-register budget), 4 none
#ifndef R3D_GATHER_PER
#define R3D_GATHER_PER 4
#endif
#ifndef R3D_GATHER_PER_BIG
#define R3D_GATHER_PER_BIG 8
#endif

namespace r3d {

// diagnostic bits of r3d_batch_t.reserved (tests force every path with them)
constexpr int kDbgSerial = 2;        // never speculate
constexpr int kDbgBands = 4;         // the depth tile in bands of at most 3 candidate rows
constexpr int kDbgDefer = 8;         // every pair is left to k_insert_big
constexpr int kDbgDropPublish = 16;  // slot 0 of scene 0 does not publish: its successors time out
constexpr int kDbgPoolTile = 32;     // every pair's depth tile and candidate list in the global pool
constexpr int kDbgNoPoolTile = 8192; // a depth tile beyond the LDS in row bands in LDS, never in the pool
constexpr int kDbgNoHits = 128;      // the kill masks from the pixel ids in global memory for every chunk (no hits kept in LDS)
constexpr int kDbgCount = 4096;      // count per pair (D_PAIRS .. D_TAKEOVER_COMMIT): two to four atomics of every pair on the same few
                                     // addresses, on the chain's critical path -- only when somebody wants to read them (bench.py's
                                     // insert_paths step, the tests)
constexpr int kDbgSparse = 16384;    // every pair's depth tile as a sparse tile (gather_sparse), whatever its size
constexpr int kDbgNoSparse = 32768;  // never: a tile beyond the LDS goes to the pool (rounds 3-5)
constexpr int kDbgVerify = 64;       // a speculative evaluation that is about to be committed is done again, now after its
                                     // predecessors, and compared (visible count, accept, visible pixels, kill masks):
                                     // counters D_VERIFY_RUNS / D_VERIFY_MISMATCH (tests/test_gpu_batch.py soaks on them)

struct ChainSlots {
  const double *samples5[kMaxChain];
  const int64_t *sample_off[kMaxChain];
  const int32_t *min_points[kMaxChain];
  const int32_t *active[kMaxChain];
  int32_t *n_visible[kMaxChain];
  int32_t *accepted[kMaxChain];
};

// published record of a finished slot
enum { REC_FLAGS = 0, REC_NTOTAL, REC_RLO, REC_RHI, REC_CLO0, REC_CHI0, REC_CLO1, REC_CHI1 };
constexpr int kRecAccepted = 1, kRecRebased = 2, kRecFar = 4;
constexpr int kProgDeferred = -2;                        // (-1: round 4's time-out mark, no longer written)
// hand-over word of a pair (BatchWs::park) and the header a parked pair leaves (BatchWs::park_hdr, kParkInts words)
constexpr int kParkNone = 0, kParkParked = 1, kParkPredDone = 2;
constexpr int kParkRaw = 0, kParkEvaluated = 1;      // PK_KIND: nothing usable was left (evaluate after the predecessors) | a record
enum { PK_KIND = 0, PK_P0, PK_NVALID, PK_NVIS, PK_ACCEPT, PK_REBASE, PK_FLAGS, PK_NKILL, PK_OFF_LO, PK_OFF_HI,
       PK_RMIN, PK_RMAX, PK_CMIN0, PK_CMIN1, PK_CMAX0, PK_CMAX1, PK_VRMIN, PK_VRMAX, PK_VCMIN0, PK_VCMIN1, PK_VCMAX0, PK_VCMAX1,
       PK_SIG_LO, PK_SIG_HI, PK_VERIFY };
static_assert(PK_VERIFY < kParkInts, "the header of a parked pair");

// ---- window of the range image: rows [r_lo, r_hi] x one or two column intervals of whole 32-pixel
// words (two when the object straddles the azimuth seam) ---------------------------------------------
// Quotient and remainder of 0 <= p < 2^24 by a divisor fixed for the workgroup: one float multiply and a
// correction step instead of the ~40 instructions of an integer division by a run-time value.  (float)p is
// exact, the reciprocal and the product round by 2^-24 each: the truncated product is within 1 of the quotient.
struct FastDiv {
  int d;
  float inv;
  __device__ __forceinline__ void set(int d_) {
    d = d_;
    inv = 1.0f / (float)(d_ > 0 ? d_ : 1);
  }
  __device__ __forceinline__ int div(int p, int &rem) const {
    int q = (int)((float)p * inv);
    int r = p - q * d;
    if (r < 0) {
      --q;
      r += d;
    } else if (r >= d) {
      ++q;
      r -= d;
    }
    rem = r;
    return q;
  }
};

struct Window {
  int r_lo, r_hi, n_iv, jl0, jh0, jl1, jh1, nj0, njw, nrw, cols;
  FastDiv by_njw;
  // window-local word index of image word (row r, word j), -1 outside the window
  __device__ __forceinline__ int lword(int r, int j) const {
    if (r < r_lo || r > r_hi) return -1;
    int k;
    if (j >= jl0 && j <= jh0) k = j - jl0;
    else if (n_iv > 1 && j >= jl1 && j <= jh1) k = nj0 + j - jl1;
    else return -1;
    return (r - r_lo) * njw + k;
  }
  __device__ __forceinline__ void row_word(int e, int &r, int &j) const {            // e: local word
    int k;
    r = r_lo + by_njw.div(e, k);
    j = k < nj0 ? jl0 + k : jl1 + (k - nj0);
  }
  __device__ __forceinline__ int lpix_rc(int r, int c) const {  // window-local pixel, -1 outside
    int lw = lword(r, c >> 5);
    return lw < 0 ? -1 : (lw << 5) + (c & 31);
  }
  __device__ __forceinline__ bool touches_words(int jmin, int jmax) const {
    return (jmin <= jh0 && jmax >= jl0) || (n_iv > 1 && jmin <= jh1 && jmax >= jl1);
  }
};

// The pixels whose scene depth an evaluation reads: the window's rows x exact column intervals.
struct DTile {
  int r0, r1, n_iv, c00, c10, c01, c11, w0, W, npx;      // interval 0: [c00, c10], interval 1: [c01, c11]
  __device__ __forceinline__ int index(int r, int c) const {
    if (r < r0 || r > r1) return -1;
    int k;
    if (c >= c00 && c <= c10) k = c - c00;
    else if (n_iv > 1 && c >= c01 && c <= c11) k = w0 + c - c01;
    else return -1;
    return (r - r0) * W + k;
  }
};

struct WinImage {
  uint32_t *w;
  __device__ __forceinline__ bool get_local(int lp) const { return (w[lp >> 5] >> (lp & 31)) & 1u; }
  __device__ __forceinline__ void set_local(int lp) { atomicOr(&w[lp >> 5], 1u << (lp & 31)); }
  __device__ __forceinline__ uint32_t word(const Window &win, int r, int j) const {
    int lw = win.lword(r, j);
    return lw < 0 ? 0u : w[lw];
  }
};

// closing.py:44-57 on up to 15 already loaded keys (R3D_SENT = empty): sum over the occupied ones,
// drow outer / dcolumn inner, divided by their count.
__device__ __forceinline__ double mean_of_keys(const unsigned long long (&v)[15]) {
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

// Diagnostic builds (make STAMPS=1) record a 100 MHz wall-clock stamp per phase in the first bytes
// of the scene's out_xyzi slab (scratch until r3d_batch_finish), 32 words per slot;
// tools/stamps_insert.py reads them.
#ifdef R3D_STAMPS
#define STAMP(i)                                                                                         \
  do {                                                                                                   \
    __syncthreads();                                                                                     \
    if (tid == 0) reinterpret_cast<long long *>(b.out_xyzi + (int64_t)s * b.cap * 4)[slot_no * 32 + (i)] = wall_clock64(); \
  } while (0)
// inside the gather loop: thread 0's own time between the marks, loads drained at every mark
#define GSTAMP_DECL long long g_acc[5] = {0, 0, 0, 0, 0}, g_last = 0
#define GSTAMP(i)                                                   \
  do {                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    long long g_now = wall_clock64();                               \
    if ((i) > 0) g_acc[i] += g_now - g_last;                        \
    g_last = g_now;                                                 \
  } while (0)
#define GSTAMP_END                                                                                              \
  do {                                                                                                          \
    if (tid == 0)                                                                                               \
      for (int gi = 1; gi < 5; ++gi)                                                                            \
        reinterpret_cast<long long *>(b.out_xyzi + (int64_t)s * b.cap * 4)[slot_no * 32 + 16 + gi] += g_acc[gi]; \
  } while (0)
#else
#define STAMP(i)
#define GSTAMP_DECL
#define GSTAMP(i)
#define GSTAMP_END
#endif

// header of a workgroup's LDS (ints)
enum {
  H_NVALID = 0, H_NCAND, H_REBASE, H_FLAGS, H_RMIN, H_RMAX, H_CMIN0, H_CMIN1, H_CMAX0, H_CMAX1,
  H_NLIST, H_CARRY, H_EXT0, H_EXT1, H_NOCC, H_NVIS, H_VRMIN, H_VRMAX, H_VCMIN0, H_VCMIN1, H_VCMAX0,
  H_VCMAX1, H_FARADD, H_FILL, H_NHIT, H_HITEND,
  H_SFAR,             // the sample has a point beyond 500 m (its commit may add to the far list: such a pair is never parked as a record)
  H_SIG = 28,         // two words: the signature's 64-bit sum (diagnostic bit 64)
  H_PHASE_END = 30,   // the cells below belong to the phases (sample_phase clears them)
  H_GO = 32,          // chain logic: broadcast cells [H_GO, H_GO + 1] (not touched by the phases)
  H_SCAN = 36         // block scan cells [NT/64 + 1]
};
static_assert(H_SFAR < H_SIG && (H_SIG & 1) == 0, "header cells");
constexpr int kHdrBytes = 512;

enum { kOk = 0, kNoFit = 1, kNeedSerial = 2, kStale = 3 };
// diagnostic counters (BatchWs::dbg; r3d_batch_debug_counters)
enum { D_POOL_FULL = 0, D_TILE_POOLED, D_EVAL_TWICE, D_VERIFY_RUNS, D_VERIFY_MISMATCH, D_HITS_OVERFLOW, D_DEFERRED, D_REBASE,
       D_REBASE_OOB, D_REBASE_HOLDER, D_REBASE_FAR, D_REBASE_OTHER,       // (a diagnostic build, -DR3D_CHECK, counts its failed checks in [12 .. 15], notes in [16 .. 31])
       // round 5, [32 ..]: pairs committed from their own evaluation and the chunks they listed; pairs left to whoever finishes
       // their predecessors -- with a record of the evaluation / as they came --; parked pairs committed from their record
       D_PAIRS = 32, D_CHUNKS_LISTED, D_PARKED, D_PARKED_RAW, D_TAKEOVER_COMMIT,
       D_VIRTUAL /* scenes put into virtual order at step 0: counted by k_virt_scan, r3d_batch.hip */,
       D_SPARSE /* round 6: evaluations on a sparse tile */, D_SPARSE_NOFIT /* ... that it could not hold either */ };
static_assert(D_VIRTUAL == kCntVirtual, "the counter r3d_batch.hip writes");
constexpr int kDbgInts = 64;

// Diagnostic builds (-DR3D_CHECK): the index of every access the gather / kill / commit code derives from data is
// checked against its array; a violation is counted in BatchWs::dbg[8 + code] and the access skipped.
#ifdef R3D_CHECK
#define CHK(cond, code) ((cond) ? true : (atomicAdd(&w.dbg[12 + ((code) & 3)], 1), false))
#else
#define CHK(cond, code) true
#endif

// A value every lane holds (read from LDS or global memory): into a scalar register.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// NT threads; POOL: the scratch images of a window too large for the LDS may live in the pool (the flavour for large
// range images: every access to those images is then a flat one, which the other avoids)
template <int NT, bool POOL>
struct Ins {
  const r3d_batch_t &b;
  const BatchWs &w;
  unsigned char *smem;
  const int lds_cap, s, chunks, slot_no;
  const int tid, rows, cols, npix, wpr;
  const bool force_glist;
  int *H, *scan;
  // the slot's candidate
  const double *rows5;
  int m, need, step;
  // sample phase results
  bool tiny_el;                           // the scene's elevation span is so small that a bound's holder may sit in any row
  Window win;
  DTile dt;
  int nvalid, ww, nocc, ncand, r1, rec_end;
  bool planes_pooled;
  uint32_t *s_oob, *s_lp, *s_img, *s_rank;
  uint16_t *s_F, *s_start;
  unsigned long long *s_sdepth;
  WinImage A, T, Cs, D, E;
  // scene phase results
  uint16_t *s_V;
  uint32_t *s_cand;
  unsigned long long *s_dtile;
  unsigned long long *g_dtile;           // the tile in the global pool (the window does not fit the LDS), or null
  uint32_t *g_cand;
  long long pool_off;                    // this pair's piece of the pool (-1: none yet, -2: the pool was exhausted)
  unsigned char *s_list, *g_list;        // chunk list: 32-byte entries growing down from the end of the LDS, or of
  bool glist;                            // the pair's area in global memory when they do not fit there
  DTile bt;                              // the band of the tile currently in LDS
  int list_cap, nlist, nvis, n_base, n_far;
  // the hits (gather_flat): {point number, tile pixel << 16 | window pixel} of every living scene point inside the
  // tile, entry by entry (the entry's `hbase` is where its in-tile points start, in lane order), in LDS behind the tile
  // as far as the room goes: the coordinates are then fetched for the hits alone, all at once, and the kill masks of
  // an accepted pair are formed without a second trip to the pixel ids
  uint2 *s_hit;
  int hit_cap;
  bool intile;                            // the gather leaves, in every list entry's kill field, which of its points lie inside the tile
  bool flat;                              // gather_flat does the gather
  bool lazy_root;                         // the pooled tile keeps squared depths (tile_key)
  // the SPARSE tile (round 6): a window whose dense tile exceeds the LDS keeps only the pixels the evaluation reads --
  // candidates the scene occupies, occupied neighbours of the candidates that are holes of the scene: bits in N, their
  // number before every window word in s_nrank, one depth each in s_ctile -- in LDS, instead of every pixel in the pool
  bool sparse;
  WinImage N;
  uint32_t *s_nrank;
  unsigned long long *s_ctile;
  int nneed;
  bool accept;
  bool cull_only;             // min_points < 0: the state a REJECTED candidate leaves (see commit)
  FastDiv by_W;

  // k_insert_big passes force_glist: its chunk lists live in the scene's global area (w.glist); a chain pair whose
  // list exceeds the LDS takes room from the launch's pool
  __device__ __forceinline__ Ins(const r3d_batch_t &b_, const BatchWs &w_, unsigned char *smem_, int lds_cap_, int s_,
                                 int chunks_, int slot_no_, bool force_glist_)
      : b(b_), w(w_), smem(smem_), lds_cap(lds_cap_), s(s_), chunks(chunks_), slot_no(slot_no_), tid(threadIdx.x),
        rows(b_.rows), cols(b_.cols), npix(b_.rows * b_.cols), wpr(b_.cols >> 5), force_glist(force_glist_) {
    H = reinterpret_cast<int *>(smem);
    scan = H + H_SCAN;
    nvalid = ww = nocc = ncand = nvis = nlist = 0;
    accept = false;
    glist = false;
    g_dtile = nullptr;
    g_cand = nullptr;
    s_hit = nullptr;
    hit_cap = 0;
    intile = flat = lazy_root = sparse = false;
    nneed = 0;
    pool_off = -1;
    g_list = force_glist_ ? w.glist + ((int64_t)s + 1) * chunks * kEntry : nullptr;   // entries grow down from the area's end
  }

  // A piece of the launch's pool (bytes: rounded up to 256): its offset, or -1 when the pool is exhausted.  Called by
  // the whole workgroup (two barriers).
  __device__ __forceinline__ long long pool_take(long long bytes) {
    const long long want = (bytes + 255) & ~255ll;
    __syncthreads();
    if (tid == 0) {
      unsigned long long o = atomicAdd(w.pool_head, (unsigned long long)want);
      H[H_FILL] = o + want <= (unsigned long long)w.pool_bytes ? (int)(o >> 8) : -1;
      if (H[H_FILL] < 0) atomicAdd(&w.dbg[D_POOL_FULL], 1);
    }
    __syncthreads();
    const int got = uni(H[H_FILL]);
    __syncthreads();
    return got < 0 ? -1ll : (long long)got << 8;
  }

  // chunk list entry i: { alive word, kill mask, chunk number, rows of its box (first | last << 16), hbase, - }
  __device__ __forceinline__ unsigned char *entry(int i) const { return (glist ? g_list : s_list) - kEntry * (i + 1); }
  __device__ __forceinline__ unsigned long long l_alive(int i) const {
    return glist ? *reinterpret_cast<const unsigned long long *>(g_list - kEntry * (i + 1))
                 : *reinterpret_cast<const unsigned long long *>(s_list - kEntry * (i + 1));
  }
  __device__ __forceinline__ unsigned long long l_kill(int i) const {
    return glist ? *reinterpret_cast<const unsigned long long *>(g_list - kEntry * (i + 1) + 8)
                 : *reinterpret_cast<const unsigned long long *>(s_list - kEntry * (i + 1) + 8);
  }
  __device__ __forceinline__ void set_kill(int i, unsigned long long m) const {
    if (glist) *reinterpret_cast<unsigned long long *>(g_list - kEntry * (i + 1) + 8) = m;
    else *reinterpret_cast<unsigned long long *>(s_list - kEntry * (i + 1) + 8) = m;
  }
  __device__ __forceinline__ uint32_t l_chunk(int i) const {
    return glist ? *reinterpret_cast<const uint32_t *>(g_list - kEntry * (i + 1) + 16)
                 : *reinterpret_cast<const uint32_t *>(s_list - kEntry * (i + 1) + 16);
  }
  __device__ __forceinline__ uint32_t l_rows(int i) const {
    return glist ? *reinterpret_cast<const uint32_t *>(g_list - kEntry * (i + 1) + 20)
                 : *reinterpret_cast<const uint32_t *>(s_list - kEntry * (i + 1) + 20);
  }
  __device__ __forceinline__ int l_hbase(int i) const {
    return glist ? *reinterpret_cast<const int *>(g_list - kEntry * (i + 1) + 24)
                 : *reinterpret_cast<const int *>(s_list - kEntry * (i + 1) + 24);
  }
  __device__ __forceinline__ void set_hbase(int i, int hb) const {
    if (glist) *reinterpret_cast<int *>(g_list - kEntry * (i + 1) + 24) = hb;
    else *reinterpret_cast<int *>(s_list - kEntry * (i + 1) + 24) = hb;
  }
  __device__ __forceinline__ void set_entry(int i, unsigned long long a, uint32_t c, uint32_t rr) const {
    if (glist) {
      *reinterpret_cast<ulonglong2 *>(g_list - kEntry * (i + 1)) = make_ulonglong2(a, 0ull);
      *reinterpret_cast<uint4 *>(g_list - kEntry * (i + 1) + 16) = make_uint4(c, rr, 0xFFFFFFFFu, 0u);
    } else {
      *reinterpret_cast<ulonglong2 *>(s_list - kEntry * (i + 1)) = make_ulonglong2(a, 0ull);
      *reinterpret_cast<uint4 *>(s_list - kEntry * (i + 1) + 16) = make_uint4(c, rr, 0xFFFFFFFFu, 0u);
    }
  }
  __device__ __forceinline__ int rank_of(int lp) const {
    return (int)s_rank[lp >> 5] + __popc(A.w[lp >> 5] & ((1u << (lp & 31)) - 1u));
  }
  __device__ __forceinline__ int global_pix(int lp) const {
    int r, j;
    win.row_word(lp >> 5, r, j);
    return pack_pix(r, (j << 5) + (lp & 31));
  }
  // Depth keys without branches (lp: window-local pixel, -1 = outside): the loads of the 15 neighbours of a
  // hole can then be in flight together -- two independent LDS reads, one dependent, instead of 15 chains.
  __device__ __forceinline__ unsigned long long sample_key(int lp) const {
    const int wd = lp >= 0 ? lp >> 5 : 0;
    const uint32_t aw = A.w[wd], rk = s_rank[wd], bit = 1u << (lp & 31);
    const bool occ = lp >= 0 && (aw & bit);
    const unsigned long long key = s_sdepth[occ ? (int)rk + __popc(aw & (bit - 1u)) : 0];
    return occ ? key : R3D_SENT;
  }
  // dl: pixel of the band in LDS / the pool (plain loads: after the gather the pooled tile is read-only, and the
  // workgroup has dropped its stale cache lines)
  // A pooled tile with many more pixels than the evaluation has candidates (`lazy_root`: a car a few metres from the sensor
  // on 448 x 2880, 100 000 pixels for a few thousand candidates) keeps the minima of the SQUARED depth the gather formed: the
  // root is taken per read instead of in a pass over the whole tile (a read-modify-write trip through L2 per pixel: config C5
  // 342 -> 311 us for such a pair).  Any other tile is converted in place (scene_phase) -- on the reference's grid the pooled
  // tiles are a few times their candidates and the evaluation reads most pixels several times (72 against 50 us with the
  // root per read).
  __device__ __forceinline__ unsigned long long tile_key(int dl) const {
    if (!g_dtile) return s_dtile[dl];
    const unsigned long long k2 = g_dtile[dl];
    if (NT != 1024) return k2;                              // (only the shape for large range images meets such tiles)
    return !lazy_root || k2 == R3D_SENT ? k2 : depth_key(sqrt(key_depth(k2)));
  }

  // the sparse tile: depth of window pixel lp (-1: outside), R3D_SENT where the tile keeps none (nobody there, or a
  // pixel the evaluation does not read)
  __device__ __forceinline__ unsigned long long ctile_key(int lp) const {
    const int wd = lp >= 0 ? lp >> 5 : 0;
    const uint32_t nw = N.w[wd], rk = s_nrank[wd], bit = 1u << (lp & 31);
    const bool has = lp >= 0 && (nw & bit);
    const unsigned long long key = s_ctile[has ? (int)rk + __popc(nw & (bit - 1u)) : 0];
    return has ? key : R3D_SENT;
  }

  // LDS layout of a pair: header | out-of-bounds bits | window pixel per point | sorted order | occupancy, closed,
  // rank images | first sorted point and min depth per occupied pixel | the three scratch images | (from r1) whatever
  // the scene phase carves.
  __device__ __forceinline__ int carve_head() {
    int carve = kHdrBytes;
    s_oob = reinterpret_cast<uint32_t *>(smem + carve);
    carve += ((m + 31) >> 5) * 4;
    s_lp = reinterpret_cast<uint32_t *>(smem + carve);
    carve += m * 4;
    s_F = reinterpret_cast<uint16_t *>(smem + carve);
    carve = (carve + m * 2 + 7) & ~7;
    return carve;
  }
  __device__ __forceinline__ int carve_images(int carve) {
    s_img = reinterpret_cast<uint32_t *>(smem + carve);
    carve = (carve + 3 * ww * 4 + 7) & ~7;
    A.w = s_img;
    Cs.w = s_img + ww;
    s_rank = s_img + 2 * ww;
    return carve;
  }
  __device__ __forceinline__ void carve_tail(int carve) {   // needs nocc
    s_start = reinterpret_cast<uint16_t *>(smem + carve);
    carve = (carve + (nocc + 1) * 2 + 7) & ~7;
    s_sdepth = reinterpret_cast<unsigned long long *>(smem + carve);
    carve += nocc * 8;
    rec_end = (carve + 15) & ~15;
    // the three scratch images behind the record -- or, for a window so large that they would leave the scene phase
    // less than its minimum (a car a few metres from the sensor on a 448 x 2880 image: 5 000 words per image), in
    // the launch's pool: place_scratch_images() below.  (Only then: the evaluation reads these bits far more often
    // than the depth tile, which goes to the pool first.)
    const long long after_sort = 4ll * nocc + 2ll * nvalid + 64;                 // the sample phase's sort scratch
    const long long after_scene = 2ll * nvalid + 44ll * dt.W + 4 * 1024;           // visible list, one band of the tile, slack
    planes_pooled = POOL && (int64_t)rec_end + 3ll * ww * 4 + (after_sort > after_scene ? after_sort : after_scene) > lds_cap;
    uint32_t *scr = reinterpret_cast<uint32_t *>(smem + rec_end);
    T.w = scr;
    D.w = scr + ww;
    E.w = scr + 2 * ww;
    r1 = planes_pooled ? rec_end : (rec_end + 3 * ww * 4 + 7) & ~7;   // scratch from here on
  }
  // kOk, or kNoFit when the pool is exhausted.  Called by the whole workgroup after carve_tail().
  __device__ __forceinline__ int place_scratch_images() {
    if (!POOL || !planes_pooled) return kOk;
    const long long off = pool_take(3ll * ww * 4);
    if (off < 0) return kNoFit;
    uint32_t *scr = reinterpret_cast<uint32_t *>(w.tile_pool + off);
    T.w = scr;
    D.w = scr + ww;
    E.w = scr + 2 * ww;
    return kOk;
  }

#ifdef R3D_CHECK
  // Diagnostic build: a checksum of the sample's record in LDS (s_oob .. s_sdepth, everything the sample phase leaves for
  // the scene phase and the commit) -- taken at the end of the sample phase (keep = true), compared at the stations of
  // the scene phase: which phase overwrites the record?  Whole workgroup; cell H_SIG + 1 holds the sum, H_SIG the work.
  __device__ __forceinline__ void record_check(int station, bool keep = false) {
    __syncthreads();
    if (tid == 0) H[H_SIG] = 0;
    __syncthreads();
    unsigned sum = 0u;
    const uint32_t *wds = reinterpret_cast<const uint32_t *>(smem);
    for (int i = kHdrBytes / 4 + tid; i < rec_end / 4; i += NT) sum += wds[i] * (uint32_t)(2 * i + 1);
    sum = (unsigned)wave_sum_i32((int)sum);
    if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned *>(&H[H_SIG]), sum);
    __syncthreads();
    // ... and do the waves agree on what every one of them keeps in scalar registers (the layout of the record, the window)?
    {
      unsigned hsh = (unsigned)nvalid * 2654435761u;
      hsh = (hsh ^ (unsigned)ww) * 2246822519u;
      hsh = (hsh ^ (unsigned)rec_end) * 3266489917u;
      hsh = (hsh ^ (unsigned)r1) * 668265263u;
      hsh = (hsh ^ (unsigned)m) * 374761393u;
      hsh = (hsh ^ (unsigned)(reinterpret_cast<unsigned char *>(A.w) - smem)) * 2654435761u;
      hsh = (hsh ^ (unsigned)(reinterpret_cast<unsigned char *>(s_F) - smem)) * 2246822519u;
      hsh = (hsh ^ (unsigned)(reinterpret_cast<unsigned char *>(s_start) - smem)) * 3266489917u;
      hsh = (hsh ^ (unsigned)dt.npx) * 668265263u;
      hsh = (hsh ^ (unsigned)win.njw) * 374761393u;
      hsh = (hsh ^ (unsigned)nocc) * 2654435761u;
      if ((tid & 63) == 0) scan[tid >> 6] = (int)hsh;
    }
    __syncthreads();
    if (tid == 0) {
      int differ = -1;
      for (int v = 1; v < NT / 64; ++v)
        if (scan[v] != scan[0]) differ = v;
      const bool sum_bad = !keep && H[H_SIG + 1] != H[H_SIG];
      if (keep) H[H_SIG + 1] = H[H_SIG];
      if ((sum_bad || differ >= 0) && atomicAdd(&w.dbg[12], 1) < 4) {
        const int slot = 16 + 4 * (atomicAdd(&w.dbg[13], 1) & 3);
        w.dbg[slot] = station | (differ >= 0 ? 0x100 * (differ + 1) : 0) | (sum_bad ? 0x10000 : 0);
        w.dbg[slot + 1] = s, w.dbg[slot + 2] = step, w.dbg[slot + 3] = rec_end;
      }
    }
    __syncthreads();
  }
#define RECORD_CHECK(st) record_check(st)
#else
#define RECORD_CHECK(st)
#endif

  // -- the window of a projected sample (H_RMIN .. H_CMAX1, H_NVALID in the header): candidates lie within 2 rows /
  // 1 column of a sample pixel, their hole means look 2 / 1 further, their closing 4 / 2 further.  Sets win, dt, ww.
  __device__ __forceinline__ void compute_window() {
    win.cols = cols;
    win.n_iv = 1;
    win.jl0 = win.jl1 = win.jh1 = 0;
    win.jh0 = -1;
    win.r_lo = 0;
    win.r_hi = -1;                                          // nothing valid: empty window
    dt.n_iv = 0;
    dt.c00 = dt.c01 = dt.c11 = 0;
    dt.c10 = -1;
    if (nvalid > 0) {
      const int rmin = uni(H[H_RMIN]), rmax = uni(H[H_RMAX]);
      const int cmin0 = uni(H[H_CMIN0]), cmax0 = uni(H[H_CMAX0]), cmin1 = uni(H[H_CMIN1]), cmax1 = uni(H[H_CMAX1]);
      win.r_lo = rmin - 6 < 0 ? 0 : rmin - 6;
      win.r_hi = rmax + 6 > rows - 1 ? rows - 1 : rmax + 6;
      const bool h0 = cmax0 >= 0, h1 = cmax1 >= 0;
      // exact column interval of either image half, and the whole words that hold it
      const int lo0 = cmin0 - 3 < 0 ? 0 : cmin0 - 3, hi0 = cmax0 + 3 > cols - 1 ? cols - 1 : cmax0 + 3;
      const int lo1 = cmin1 - 3 < 0 ? 0 : cmin1 - 3, hi1 = cmax1 + 3 > cols - 1 ? cols - 1 : cmax1 + 3;
      if (h0 && h1) {
        const bool merge_w = (lo1 >> 5) <= (hi0 >> 5) + 1, merge_c = lo1 <= hi0 + 1;
        win.n_iv = merge_w ? 1 : 2;
        win.jl0 = lo0 >> 5;
        win.jh0 = merge_w ? ((hi1 >> 5) > (hi0 >> 5) ? (hi1 >> 5) : (hi0 >> 5)) : (hi0 >> 5);
        win.jl1 = merge_w ? 0 : (lo1 >> 5);
        win.jh1 = merge_w ? 0 : (hi1 >> 5);
        dt.n_iv = merge_c ? 1 : 2;
        dt.c00 = lo0;
        dt.c10 = merge_c ? (hi1 > hi0 ? hi1 : hi0) : hi0;
        dt.c01 = merge_c ? 0 : lo1;
        dt.c11 = merge_c ? 0 : hi1;
      } else {
        const int lo = h0 ? lo0 : lo1, hi = h0 ? hi0 : hi1;
        win.jl0 = lo >> 5;
        win.jh0 = hi >> 5;
        dt.n_iv = 1;
        dt.c00 = lo;
        dt.c10 = hi;
      }
  }
  win.nj0 = win.jh0 - win.jl0 + 1;
  win.njw = win.nj0 + (win.n_iv > 1 ? win.jh1 - win.jl1 + 1 : 0);
  win.nrw = win.r_hi - win.r_lo + 1;
  win.by_njw.set(win.njw);
  ww = win.nrw * win.njw;                                 // window words
  dt.r0 = win.r_lo;
  dt.r1 = win.r_hi;
  dt.w0 = dt.c10 - dt.c00 + 1;
  dt.W = dt.n_iv == 0 ? 0 : dt.w0 + (dt.n_iv > 1 ? dt.c11 - dt.c01 + 1 : 0);
  dt.npx = win.nrw * dt.W;
  by_W.set(dt.W);
  }

  // ================================================================================================
  // sample phase.  kOk, or kNoFit when the per-point / per-pixel arrays exceed this kernel's LDS.
  // ================================================================================================
  __device__ __forceinline__ int sample_phase() {
    accept = false;
    nvis = 0;
    int carve = carve_head();
    if (carve > lds_cap) return kNoFit;
    __syncthreads();                                         // the previous use of this LDS is over
    for (int i = tid; i < ((m + 31) >> 5); i += NT) s_oob[i] = 0u;
    if (tid < H_PHASE_END)
      H[tid] = (tid == H_RMIN || tid == H_CMIN0 || tid == H_CMIN1 || tid == H_VRMIN || tid == H_VCMIN0 || tid == H_VCMIN1)
                   ? 0x7FFFFFFF
                   : (tid == H_RMAX || tid == H_CMAX0 || tid == H_CMAX1 || tid == H_EXT0 || tid == H_EXT1 ||
                      tid == H_VRMAX || tid == H_VCMAX0 || tid == H_VCMAX1)
                         ? -1
                         : 0;
    __syncthreads();
    const Binning bn = make_binning(b.bounds[2 * s + 0], b.bounds[2 * s + 1], rows, cols);
    tiny_el = bn.d_el < 1e-4;

    STAMP(0);
    // -- 1. project the sample with the scene's bounds, sample=True (insertion.py:455-459) ---------
    {
      // column ranges are kept per image half so that an object across the azimuth seam (columns
      // 0 and cols-1) yields two narrow windows instead of one full-width window
      const int half = cols >> 1;
      int rmin = 0x7FFFFFFF, rmax = -1, cmin0 = 0x7FFFFFFF, cmax0 = -1, cmin1 = 0x7FFFFFFF, cmax1 = -1;
      int nval = 0, flags = 0;
      // The bin of a sample point is guessed in float32 and confirmed in float64 on the edges of that bin, as step 0 does
      // for the scene's points (confirm_bin, r3d_batch.hpp; row table of k_prepare / the last rebase).  A confirmed bin
      // in rows 1 .. rows-2 also says that the elevation lies strictly inside the scene's bounds; everything else --
      // unconfirmed, first or last row, outside the image, not finite -- takes the reference formula (spherical_bin).
      const double *row_cc = w.row_q + (int64_t)s * (rows + 2);
      const float inv_del = (float)(1.0 / bn.d_el), inv_daz = (float)(1.0 / bn.d_az), elo = (float)(bn.min_el + 0.00001);
      for (int j = tid; j < m; j += NT) {
        uint32_t key = 0xFFFFFFFFu;
        const double *q = rows5 + (int64_t)j * 5;
        const double x = q[0], y = q[1], z = q[2];
        int row, col, ok;
        {
          const float fx = (float)x, fy = (float)y, fz = (float)z;
          const float ssf = fmaf(fx, fx, fmaf(fy, fy, fz * fz));
          const float qf = __builtin_amdgcn_fmed3f(fz * __frsqrt_rn(ssf), -1.f, 1.f);
          row = (int)floorf((guess_acosf(qf) - elo) * inv_del);
          col = (int)((guess_atan2f(fy, fx) + 3.14159274f) * inv_daz);
          row = max(1, min(row, rows - 2));
          col = max(0, min(col, cols - 1));
          ok = rows > 2 && confirm_bin(row_cc, w.col_dir, row, col, x, y, z, x * x + y * y + z * z) ? 7 : 0;
        }
        if (!ok) {
          SphBin sb = spherical_bin(bn.max_el, bn.min_el, rows, cols, x, y, z);
          row = sb.row, col = sb.col, ok = sb.ok;
          if (ok & 16) atomicAdd(&w.dbg[kCntEdgeSample], 1);        // (within 1e-12 of a bin edge: counted, r3d_batch.hpp)
        }
        if (!(ok & 4)) {
          flags |= R3D_S_NONFINITE;
        } else if (ok & 1) {                         // rows outside [0, rows) are skipped (:107-108)
          if (!(ok & 2)) {
            flags |= R3D_S_COL_RANGE;                // assert :112
          } else {
            key = ((uint32_t)row << 16) | (uint32_t)col;    // re-keyed by window pixel below
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
            if (ok & 8) atomicOr(&s_oob[j >> 5], 1u << (j & 31));
          }
        }
        s_lp[j] = key;
      }
      nval = wave_sum_i32(nval);
      flags = wave_or_i32(flags);
      rmin = wave_min_i32(rmin); rmax = wave_max_i32(rmax);
      cmin0 = wave_min_i32(cmin0); cmax0 = wave_max_i32(cmax0);
      cmin1 = wave_min_i32(cmin1); cmax1 = wave_max_i32(cmax1);
      if ((tid & 63) == 0) {
        atomicAdd(&H[H_NVALID], nval);
        if (flags) atomicOr(&H[H_FLAGS], flags);
        atomicMin(&H[H_RMIN], rmin);
        atomicMax(&H[H_RMAX], rmax);
        atomicMin(&H[H_CMIN0], cmin0);
        atomicMax(&H[H_CMAX0], cmax0);
        atomicMin(&H[H_CMIN1], cmin1);
        atomicMax(&H[H_CMAX1], cmax1);
      }
    }
    __syncthreads();
    nvalid = uni(H[H_NVALID]);

    STAMP(1);
    compute_window();

    // bit images of the sample: occupancy | closed | occupied sample pixels before each window word.  (The three
    // scratch images -- dilations / visible pixels, scene occupancy, scene closed -- follow the sample's record.)
    if ((int64_t)carve + 3ll * ww * 4 + 64 > lds_cap) return kNoFit;
    carve = carve_images(carve);
    for (int i = tid; i < ww; i += NT) A.w[i] = 0u;
    // every valid sample pixel lies inside the window
    for (int j = tid; j < m; j += NT) {
      uint32_t rc = s_lp[j];
      if (rc != 0xFFFFFFFFu) s_lp[j] = (uint32_t)win.lpix_rc((int)(rc >> 16), (int)(rc & 0xFFFF));
    }
    __syncthreads();

    STAMP(2);
    // -- 3. sample occupancy, rank of every occupied sample pixel ------------------------------------
    for (int j = tid; j < m; j += NT) {
      uint32_t lp = s_lp[j];
      if (lp != 0xFFFFFFFFu) A.set_local((int)lp);
    }
    __syncthreads();
    for (int base = 0; base < ww; base += NT) {            // exclusive prefix popcount over the words
      int e = base + tid;
      int c = e < ww ? __popc(A.w[e]) : 0;
      int tot;
      int ex = block_escan_i32(c, scan, tot);
      int carry0 = H[H_CARRY];
      if (e < ww) s_rank[e] = (uint32_t)(carry0 + ex);
      __syncthreads();
      if (tid == 0) H[H_CARRY] = carry0 + tot;
      __syncthreads();
    }
    nocc = uni(H[H_CARRY]);
    // EVERY wave has read the carry before thread 0 reuses the cell further down: without this barrier a wave that the CU
    // schedules late (other workgroups, other kernels in flight) finds the cell already zeroed, lays the record out for
    // nocc = 0 and writes over what the others have built -- once per several million pairs, and only under load: wrong
    // visible lists, lost chunks, a spurious rebase, now and then an access far outside (rounds 2-4)
    __syncthreads();
    if (tid == 0) H[H_NOCC] = nocc;

    // per occupied pixel: first sorted point, min depth; scratch: counters, unordered placement
    carve_tail(carve);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + r1);
    uint16_t *s_U = reinterpret_cast<uint16_t *>(smem + r1 + nocc * 4);
    if ((int64_t)r1 + (int64_t)nocc * 4 + (int64_t)nvalid * 2 > lds_cap) return kNoFit;
    if (place_scratch_images() != kOk) return kNoFit;
    for (int i = tid; i < nocc; i += NT) {
      s_cnt[i] = 0u;
      s_sdepth[i] = R3D_SENT;
    }
    if (tid == 0) H[H_CARRY] = 0;
    __syncthreads();

    STAMP(3);
    // -- 4. counting sort by (pixel, sample index): the order of visible_sample (insertion.py:474-482)
    for (int j = tid; j < m; j += NT) {
      uint32_t lp = s_lp[j];
      if (lp != 0xFFFFFFFFu) atomicAdd(&s_cnt[rank_of((int)lp)], 1u);
    }
    __syncthreads();
    for (int base = 0; base < nocc; base += NT) {
      int e = base + tid;
      int c = e < nocc ? (int)s_cnt[e] : 0;
      int tot;
      int ex = block_escan_i32(c, scan, tot);
      int carry0 = H[H_CARRY];
      if (e < nocc) {
        s_start[e] = (uint16_t)(carry0 + ex);
        s_cnt[e] = (uint32_t)(carry0 + ex);                 // the pixel's cursor
      }
      __syncthreads();
      if (tid == 0) H[H_CARRY] = carry0 + tot;
      __syncthreads();
    }
    if (tid == 0) s_start[nocc] = (uint16_t)nvalid;
    for (int j = tid; j < m; j += NT) {
      uint32_t lp = s_lp[j];
      if (lp != 0xFFFFFFFFu) s_U[atomicAdd(&s_cnt[rank_of((int)lp)], 1u)] = (uint16_t)j;
    }
    __syncthreads();
    // a point's place inside its pixel's run = how many points of the run have a smaller index;
    // depth of the pixel = min r over its points (insertion.py:118-125)
    for (int j = tid; j < m; j += NT) {
      uint32_t lp = s_lp[j];
      if (lp == 0xFFFFFFFFu) continue;
      int rk = rank_of((int)lp);
      int a = s_start[rk], z = s_start[rk + 1], before = 0;
      for (int p = a; p < z; ++p) before += (int)s_U[p] < j ? 1 : 0;
      s_F[a + before] = (uint16_t)j;
      const double *q = rows5 + (int64_t)j * 5;
      double x = q[0], y = q[1], zc = q[2];
      const double rr = sqrt(x * x + y * y + zc * zc);
      unsigned long long key = depth_key(rr);
      atomicMin(&s_sdepth[rk], key);
      if (rr > R3D_EMPTY_DEPTH) H[H_SFAR] = 1;                 // (every writer writes 1)
    }
    __syncthreads();

    STAMP(4);
    // -- 5. closing of the sample's occupancy (closing.py:9-23) by word-parallel dilate / erode; exact
    // on every row at least 2 inside the window (or at the image border): candidates are ------------
    closing(A, T, Cs);
    // candidate pixels: where the sample is closed
    {
      int c = 0;
      for (int e = tid; e < ww; e += NT) c += __popc(Cs.w[e]);
      c = wave_sum_i32(c);
      if ((tid & 63) == 0 && c) atomicAdd(&H[H_NCAND], c);
    }
    __syncthreads();
    ncand = uni(H[H_NCAND]);
    STAMP(5);
#ifdef R3D_CHECK
    record_check(0, true);
#endif
    return kOk;
  }

  // Tile pixel (one band = the whole window) and window pixel of a packed pixel id in one go: both number the window's
  // rows from r_lo, the tile its exact columns, the window the 32-pixel words that hold them.  -1: outside.
  __device__ __forceinline__ void place_rc(uint32_t p, int &dl, int &lp) const {
    const int nj1 = win.n_iv > 1 ? win.jh1 - win.jl1 + 1 : 0, w1 = dt.n_iv > 1 ? dt.c11 - dt.c01 + 1 : 0;
    const int rr = pix_row(p) - win.r_lo, c = pix_col(p), j = c >> 5;
    const int k0 = j - win.jl0, k1 = j - win.jl1;
    const bool in0 = (unsigned)k0 < (unsigned)win.nj0, in1 = (unsigned)k1 < (unsigned)nj1;
    const int t0 = c - dt.c00, t1 = c - dt.c01;
    const bool ok_r = (unsigned)rr < (unsigned)win.nrw;
    const bool tin0 = (unsigned)t0 < (unsigned)dt.w0, tin1 = (unsigned)t1 < (unsigned)w1;   // (w1 = 0: one interval)
    lp = ok_r && (in0 || in1) ? rr * (win.njw << 5) + ((in0 ? k0 : win.nj0 + k1) << 5) + (c & 31) : -1;
    dl = ok_r && (tin0 || tin1) ? rr * dt.W + (tin0 ? t0 : dt.w0 + t1) : -1;
  }

  // The chunks that can hold a point of the window: bounding box touches it, somebody alive (4 chunks
  // per thread in flight).  With super-boxes (sup != nullptr: `nsup` of them reach the window's rows, their numbers in
  // `sup`) only the chunks of those are looked at.
  __device__ __forceinline__ void build_list(const unsigned long long *boxes, const unsigned long long *alive, int n_chunks,
                                             const uint16_t *sup = nullptr, int nsup = 0) {
    constexpr int kU = 4;
    const int n_items = sup ? nsup << 6 : n_chunks;
    for (int c0 = tid; c0 < n_items; c0 += kU * NT) {
      unsigned long long bx[kU], aw[kU];
      int cc[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        int c = c0 + u * NT;
        if (sup) c = c < n_items ? ((int)sup[c >> 6] << 6) + (c & 63) : n_chunks;
        cc[u] = c;
        bx[u] = c < n_chunks ? boxes[c] : 0xFFFFull;          // empty box
        aw[u] = c < n_chunks ? __hip_atomic_load(&alive[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        int c = cc[u];
        int left = n_base - (c << 6);                       // points of the chunk below the base count
        unsigned long long a = aw[u];
        if (left < 64) a = left > 0 ? a & ((1ull << left) - 1ull) : 0ull;
        int rmin = (int)(bx[u] & 0xFFFF), rmax = (int)((bx[u] >> 16) & 0xFFFF);
        // (the exact columns of the tile, not the 32-pixel words around them: nothing outside the tile is ever read --
        // depth, occupancy and the culled points all lie inside it -- and a chunk of a ring-ordered scan is ~50 columns
        // wide, so whole words list twice the chunks)
        int cmin = (int)((bx[u] >> 32) & 0xFFFF), cmax = (int)((bx[u] >> 48) & 0xFFFF);
        bool hit = a && rmin <= win.r_hi && rmax >= win.r_lo &&
                   (box_touches_cols(cmin, cmax, dt.c00, dt.c10) || (dt.n_iv > 1 && box_touches_cols(cmin, cmax, dt.c01, dt.c11)));
        if (hit) {
          int slot = atomicAdd(&H[H_NLIST], 1);
          if (slot < list_cap) set_entry(slot, a, (uint32_t)c, (uint32_t)rmin | ((uint32_t)rmax << 16));
        }
      }
    }
  }

  // The super-boxes (64 chunks each) whose rows reach the window's: their numbers into `sup`, the count through H_CARRY.
  __device__ __forceinline__ void list_supers(uint16_t *sup, int n_chunks) {
    const int n_sup_all = (chunks + 63) >> 6, n_sup = (n_chunks + 63) >> 6;
    const int2 *rows2 = reinterpret_cast<const int2 *>(w.super_rows) + (int64_t)s * n_sup_all;
    for (int sp = tid; sp < n_sup; sp += NT) {
      const int2 r = rows2[sp];
      if (r.x <= win.r_hi && r.y >= win.r_lo) sup[atomicAdd(&H[H_CARRY], 1)] = (uint16_t)sp;
    }
  }

  // Set bit lp (< 0: none) of an image for every lane of the wave with one atomic per run of lanes that share a word.
  // The whole wave must call it.
  __device__ __forceinline__ void or_bits_by_runs(WinImage &img, int lp) {
    const int lane = tid & 63;
    const uint32_t word = lp >= 0 ? (uint32_t)(lp >> 5) : 0xFFFFFFFFu;
    uint32_t bits = lp >= 0 ? 1u << (lp & 31) : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t w2 = (uint32_t)__shfl_up((int)word, d, 64), b2 = (uint32_t)__shfl_up((int)bits, d, 64);
      if (lane >= d && w2 == word) bits |= b2;                // (bits of the same word only: a hop into an earlier run of
    }                                                         // that word adds nothing wrong)
    const uint32_t next = (uint32_t)__shfl_down((int)word, 1, 64);
    if (lp >= 0 && (lane == 63 || next != word)) atomicOr(&img.w[word], bits);
  }

  // The living points of the listed chunks whose pixel lies in rows [bt.r0, bt.r1] of the tile are min-reduced into
  // the band in LDS (insertion.py:118-125) -- on the SQUARE of the depth, x*x + y*y + z*z in the reference's order:
  // the square root is monotone, so the minimum of the roots is the root of the minimum, taken once per occupied
  // pixel when the band is complete (finish_band) instead of once per point.  A wave takes one listed chunk per step
  // (64 consecutive points: one coalesced load of pixel ids), kPer chunks in flight; coordinates are loaded only for
  // the points inside the band.  all_rows_bits: also set the scene occupancy bit of every point of the window (banded
  // and pooled tiles: the occupancy of the whole window is needed up front).
  __device__ __forceinline__ void gather(bool all_rows_bits, const uint16_t *sub, int nsub) {
    constexpr int kPer = NT == 1024 ? R3D_GATHER_PER_BIG : R3D_GATHER_PER;    // (one workgroup per CU: nothing else hides the loads)
    const int n_head = uni(b.n_head[s]), n_virt = uni(w.n_virt[s]);
    const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
    const float4 *xyzi = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    const int nitems = (sub ? nsub : nlist) << 6;           // sub: the entries whose rows reach the band
    const int lane = tid & 63;
    GSTAMP_DECL;
    for (int e0 = tid; e0 < nitems; e0 += kPer * NT) {
      int idx[kPer], dl[kPer];
      uint32_t p[kPer];
      GSTAMP(0);
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        int e = e0 + u * NT;
        idx[u] = -1;
        if (e < nitems) {
          int ent = sub ? (int)sub[e >> 6] : (e >> 6);
          if ((l_alive(ent) >> (e & 63)) & 1ull) idx[u] = (int)(l_chunk(ent) << 6) + (e & 63);
        }
      }
#pragma unroll
      for (int u = 0; u < kPer; ++u) p[u] = idx[u] >= 0 ? pixs[idx[u]] : 0u;
      GSTAMP(1);
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        dl[u] = -1;
        int lp = -1;
        if (idx[u] >= 0) {
          if (bt.npx == dt.npx) {                            // one band: the whole window
            place_rc(p[u], dl[u], lp);
            if (!(all_rows_bits && dl[u] >= 0)) lp = -1;
          } else {
            const int r = pix_row(p[u]), c = pix_col(p[u]);
            dl[u] = bt.index(r, c);
            lp = all_rows_bits && dt.index(r, c) >= 0 ? win.lpix_rc(r, c) : -1;
          }
        }
        // the scene's occupancy bit.  Images in the pool (the POOL flavour's largest windows): the 64 points of a chunk of a
        // scan in ring order fall into two or three words, and 64 atomics of one wave on the same word in L2 take their
        // turns -- the gather of a 146 000-pixel window spent 0.9 of its 1.0 ms there.  The lanes of a run of equal words
        // OR their bits together first, the last lane of the run sends one atomic.
        if (POOL && planes_pooled) {
          if (all_rows_bits) or_bits_by_runs(D, lp);
        } else if (lp >= 0) {
          D.set_local(lp);
        }
        // one band for the whole window: which points of every listed chunk lie inside the tile (the kill masks are
        // computed from those alone, and not at all for a chunk that has none)
        if (intile && !sub) {
          const unsigned long long msk = __ballot(dl[u] >= 0);
          const int e = e0 + u * NT;
          if (lane == 0 && e < nitems) set_kill(e >> 6, msk);
        }
      }
      GSTAMP(2);
      // coordinates: four float32 points in flight at a time
#pragma unroll
      for (int h = 0; h < kPer; h += 4) {
        float4 f[4];
        // (a scene in virtual order: the point of the slabs behind the listed point number -- one more dependent load,
        // for such scenes only)
        if (n_virt)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (dl[h + u] >= 0) idx[h + u] = orig_of(w, b, s, n_virt, idx[h + u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f[u] = make_float4(1.f, 0.f, 0.f, 0.f);
          if (dl[h + u] >= 0 && idx[h + u] < n_head) f[u] = xyzi[idx[h + u]];
        }
        GSTAMP(3);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (dl[h + u] < 0) continue;
          double x = (double)f[u].x, y = (double)f[u].y, z = (double)f[u].z;
          if (idx[h + u] >= n_head) load_point(b, s, idx[h + u], n_head, x, y, z);   // an inserted point: float64, from the log
          const unsigned long long key = depth_key(x * x + y * y + z * z);
          if (g_dtile) atomicMin(&g_dtile[dl[h + u]], key);
          else atomicMin(&s_dtile[dl[h + u]], key);
        }
        GSTAMP(4);
      }
    }
    GSTAMP_END;
  }

  // The same for one band that holds the whole window (512 threads and fewer: the shapes for range images of the
  // reference's size), in two flat passes instead of rounds of list -> pixel id -> coordinates: (1) every thread requests
  // the pixel ids of up to kU listed chunks at once, places them in the tile and leaves the points inside it as HITS in
  // LDS -- one reservation per wave and round --; (2) the coordinates of the hits, all in flight together.  Two dependent
  // trips to memory however long the list is.  A wave whose hits do not fit the room fetches its coordinates right away.
  __device__ __forceinline__ void gather_flat(bool all_rows_bits) {
    constexpr int kU = 8;
    const int n_head = uni(b.n_head[s]), n_virt = uni(w.n_virt[s]);
    const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
    const float4 *xyzi = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    const int nitems = nlist << 6;
    const int lane = tid & 63;
    GSTAMP_DECL;
    for (int e00 = 0; e00 < nitems; e00 += kU * NT) {
      int idx[kU];
      uint32_t p[kU], code[kU];
      unsigned long long msk[kU];
      GSTAMP(0);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int e = e00 + u * NT + tid;
        idx[u] = -1;
        if (e < nitems && ((l_alive(e >> 6) >> (e & 63)) & 1ull)) idx[u] = (int)(l_chunk(e >> 6) << 6) + (e & 63);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) p[u] = idx[u] >= 0 && CHK(idx[u] < n_base && idx[u] < b.cap, 0) ? pixs[idx[u]] : 0u;
      GSTAMP(1);
      int cnt = 0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        int dl = -1, lp = -1;
        if (idx[u] >= 0) place_rc(p[u], dl, lp);
        if (all_rows_bits && dl >= 0) D.set_local(lp);
        if (dl >= 0 && !CHK(dl < dt.npx && lp >= 0 && lp < (ww << 5), 1)) dl = -1;
        code[u] = dl >= 0 ? ((uint32_t)dl << 16) | (uint32_t)lp : 0xFFFFFFFFu;
        msk[u] = __ballot(dl >= 0);
        cnt += __popcll(msk[u]);
      }
      // room for this wave's hits of the round: one LDS atomic
      int base = -1;
      if (cnt && hit_cap > 0) {
        if (lane == 0) {
          base = atomicAdd(&H[H_NHIT], cnt);
          if (base + cnt > hit_cap) {
            atomicMin(&H[H_HITEND], base);                     // the hits end here: every later reservation lies beyond
            atomicAdd(&w.dbg[D_HITS_OVERFLOW], 1);
            base = -1;
          }
        }
        base = __builtin_amdgcn_readfirstlane(base);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int e = e00 + u * NT + tid;
        if (e < nitems) {                                      // (wave-uniform: the listed items come in 64s)
          if (lane == 0) {
            set_kill(e >> 6, msk[u]);
            set_hbase(e >> 6, base);
          }
          if (base >= 0) {
            if (code[u] != 0xFFFFFFFFu) {
              const int at = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(msk[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)msk[u], 0u));
              if (CHK(at >= 0 && at < hit_cap, 2)) s_hit[at] = make_uint2((uint32_t)idx[u], code[u]);
            }
            base += __popcll(msk[u]);
          }
        }
      }
      GSTAMP(2);
      if (base < 0 && cnt) {                                   // no room: this wave's coordinates right away
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (code[u] == 0xFFFFFFFFu) continue;
          double x, y, z;
          load_point(b, s, orig_of(w, b, s, n_virt, idx[u]), n_head, x, y, z);
          const unsigned long long key = depth_key(x * x + y * y + z * z);
          if (g_dtile) atomicMin(&g_dtile[code[u] >> 16], key);
          else atomicMin(&s_dtile[code[u] >> 16], key);
        }
      }
    }
    __syncthreads();
    GSTAMP(3);
    const int nh = uni(H[H_NHIT] < H[H_HITEND] ? H[H_NHIT] : H[H_HITEND]);
    for (int h0 = tid; h0 < nh; h0 += 4 * NT) {
      uint2 hv[4];
      float4 f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int h = h0 + u * NT;
        hv[u] = h < nh ? s_hit[h] : make_uint2(0xFFFFFFFFu, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (hv[u].x != 0xFFFFFFFFu && !CHK((int)hv[u].x < n_base && (int)(hv[u].y >> 16) < dt.npx, 3)) hv[u].x = 0xFFFFFFFFu;
      if (n_virt)                                              // (virtual order: the point of the slabs behind the hit)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (hv[u].x != 0xFFFFFFFFu) hv[u].x = (uint32_t)orig_of(w, b, s, n_virt, (int)hv[u].x);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f[u] = make_float4(1.f, 0.f, 0.f, 0.f);
        if (hv[u].x != 0xFFFFFFFFu && (int)hv[u].x < n_head) f[u] = xyzi[hv[u].x];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (hv[u].x == 0xFFFFFFFFu) continue;
        double x = (double)f[u].x, y = (double)f[u].y, z = (double)f[u].z;
        if ((int)hv[u].x >= n_head) load_point(b, s, (int)hv[u].x, n_head, x, y, z);   // an inserted point: float64, from the log
        const unsigned long long key = depth_key(x * x + y * y + z * z);
        if (g_dtile) atomicMin(&g_dtile[hv[u].y >> 16], key);
        else atomicMin(&s_dtile[hv[u].y >> 16], key);
      }
    }
    GSTAMP(4);
    GSTAMP_END;
  }

  // 5-row x 3-column closing (closing.py:9-23) of one bit image of the window: src -> tmp -> dst, dilation
  // then erosion, 32 pixels per word.  In the window-local numbering the word above / below is njw words
  // away and the horizontal neighbours are e - 1 / e + 1 unless the row (or the column interval) ends there.
  // A word outside the window reads as 0; a row or column outside the IMAGE does not take part (no
  // contribution to the dilation, no constraint on the erosion).
  __device__ __forceinline__ void closing(const WinImage &src, WinImage &tmp, WinImage &dst) {
    morph(src.w, tmp.w, 0);
    morph(tmp.w, dst.w, 1);
  }
  // pass 0: dilation, pass 1: erosion with the 5-row x 3-column element; ends with a barrier
  __device__ __forceinline__ void morph(const uint32_t *from, uint32_t *to, const int pass) {
    for (int e = tid; e < ww; e += NT) {
      int k, r = win.r_lo + win.by_njw.div(e, k);
      const int j = k < win.nj0 ? win.jl0 + k : win.jl1 + (k - win.nj0);
      const bool has_l = k > 0 && k != win.nj0, has_r = k < win.njw - 1 && k != win.nj0 - 1;
      // what stands in for a neighbour word that is not in the window: nothing, except beyond the image's
      // first / last column during the erosion
      const uint32_t l_out = pass && j == 0 ? 1u : 0u, r_out = pass && j == wpr - 1 ? 0x80000000u : 0u;
      uint32_t acc = pass ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int dr = -2; dr <= 2; ++dr) {
        const int rr = r + dr;
        if (rr < 0 || rr >= rows) continue;
        uint32_t c = 0u, l = l_out, rw = r_out;
        if (rr >= win.r_lo && rr <= win.r_hi) {
          const int q = e + dr * win.njw;
          c = from[q];
          if (has_l) l = from[q - 1] >> 31;
          if (has_r) rw = from[q + 1] << 31;
        }
        if (pass) acc &= c & ((c << 1) | l) & ((c >> 1) | rw);
        else acc |= c | (c << 1) | l | (c >> 1) | rw;
      }
      to[e] = acc;
    }
    __syncthreads();
  }

  // -- 9. visibility on the candidates: smoothed sample depth < smoothed scene depth (:461-467).  `nc` candidates in s_cand /
  // g_cand; the scene's depths from the band in LDS / the pool (bt), or -- SP -- from the sparse tile.  Visible pixels: bits
  // in T; their count, box and the rebase flag in the header.
  template <bool SP>
  __device__ __forceinline__ void evaluate(int nc) {
    WinImage &vis = T;
    const int W = dt.W;
    const int half = cols >> 1;
    int v_n = 0, v_rmin = 0x7FFFFFFF, v_rmax = -1, v_cmin0 = 0x7FFFFFFF, v_cmax0 = -1, v_cmin1 = 0x7FFFFFFF, v_cmax1 = -1;
    for (int ci = tid; ci < nc; ci += NT) {
      int lp = (int)(g_cand ? g_cand[ci] : s_cand[ci]);
      int r, c;
      win.row_word(lp >> 5, r, c);
      c = (c << 5) + (lp & 31);
      double sd = R3D_EMPTY_DEPTH, cd = R3D_EMPTY_DEPTH;
      const bool a = A.get_local(lp), d = D.get_local(lp);
      const bool c_hole = !d && E.get_local(lp);
      // A candidate lies at least 4 rows / 2 columns inside the window unless the image ends there, so the
      // neighbours of its 5 x 3 footprint are plain offsets in the window-local and tile-local numbering
      // (rows `rstride` resp. W apart); what leaves the image counts as empty.
      const int rstride = win.njw << 5, dl0 = SP ? 0 : bt.index(r, c);
      if (a) sd = key_depth(sample_key(lp));
      if (d) cd = key_depth(SP ? ctile_key(lp) : (dl0 < 0 ? R3D_SENT : tile_key(dl0)));
      // hole means (closing.py:44-57): the 15 neighbour keys are gathered first, one image at a time
      // (one register array, every load issued before the first is used), then summed in the reference's order
      if (!a) {                                            // a candidate is closed: a hole of the sample
        unsigned long long v[15];
#pragma unroll
        for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
          for (int dc = -1; dc <= 1; ++dc) {
            int rr = r + dr, cc = c + dc;
            const bool in = rr >= 0 && rr < rows && cc >= 0 && cc < cols;
            int lq = lp + dr * rstride + dc;
            v[(dr + 2) * 3 + (dc + 1)] = sample_key(in && (unsigned)lq < (unsigned)(ww << 5) ? lq : -1);
          }
        sd = mean_of_keys(v);
      }
      if (c_hole) {
        unsigned long long v[15];
#pragma unroll
        for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
          for (int dc = -1; dc <= 1; ++dc) {
            int rr = r + dr, cc = c + dc;
            const bool in = rr >= 0 && rr < rows && cc >= 0 && cc < cols;
            if (SP) {                                      // (the window's numbering, as for the sample's holes above)
              int lq = lp + dr * rstride + dc;
              v[(dr + 2) * 3 + (dc + 1)] = ctile_key(in && (unsigned)lq < (unsigned)(ww << 5) ? lq : -1);
            } else {
              int dq = dl0 + dr * W + dc;
              const bool ok = in && dl0 >= 0 && (unsigned)dq < (unsigned)bt.npx;
              unsigned long long key = tile_key(ok ? dq : 0);
              v[(dr + 2) * 3 + (dc + 1)] = ok ? key : R3D_SENT;
            }
          }
        cd = mean_of_keys(v);
      }
      if (sd < cd) {
        vis.set_local(lp);
        v_rmin = r < v_rmin ? r : v_rmin;
        v_rmax = r > v_rmax ? r : v_rmax;
        if (c < half) {
          v_cmin0 = c < v_cmin0 ? c : v_cmin0;
          v_cmax0 = c > v_cmax0 ? c : v_cmax0;
        } else {
          v_cmin1 = c < v_cmin1 ? c : v_cmin1;
          v_cmax1 = c > v_cmax1 ? c : v_cmax1;
        }
        if (a) {                                           // its sample points are visible (:474)
          int rk = rank_of(lp);
          int p0 = s_start[rk], p1 = s_start[rk + 1];
          v_n += p1 - p0;
          for (int p = p0; p < p1; ++p) {
            int j = s_F[p];
            if (!cull_only && ((s_oob[j >> 5] >> (j & 31)) & 1u)) atomicOr(&H[H_REBASE], 1);   // bounds move: new extreme elevation
          }
        }
      }
    }
    v_n = wave_sum_i32(v_n);
    v_rmin = wave_min_i32(v_rmin); v_rmax = wave_max_i32(v_rmax);
    v_cmin0 = wave_min_i32(v_cmin0); v_cmax0 = wave_max_i32(v_cmax0);
    v_cmin1 = wave_min_i32(v_cmin1); v_cmax1 = wave_max_i32(v_cmax1);
    if ((tid & 63) == 0 && v_rmax >= 0) {
      atomicAdd(&H[H_NVIS], v_n);
      atomicMin(&H[H_VRMIN], v_rmin);
      atomicMax(&H[H_VRMAX], v_rmax);
      atomicMin(&H[H_VCMIN0], v_cmin0);
      atomicMax(&H[H_VCMAX0], v_cmax0);
      atomicMin(&H[H_VCMIN1], v_cmin1);
      atomicMax(&H[H_VCMAX1], v_cmax1);
    }
  }

  // ---- the sparse tile (round 6) ----------------------------------------------------------------------------------------
  // A window whose dense depth tile exceeds the LDS (a car a few metres away: 15 000 pixels on the reference's grid, 100 000
  // on 448 x 2880) used to get its tile in the launch's pool in global memory: cleared pixel by pixel, filled with global
  // atomics -- 37-47 G/s on this part, profiles/r06_image_build.md --, square-rooted pixel by pixel, read through L2; for an
  // evaluation that reads a few thousand of those pixels.  Now: pass 1 over the listed chunks sets the scene's occupancy bits
  // (pixel ids only); the closing says where the scene has holes; the pixels the evaluation can read are the candidates
  // the scene occupies and the occupied 5 x 3 neighbours of the candidates that are holes of the scene (bits N, ranked);
  // pass 2 fetches coordinates for the points in THOSE pixels only and min-reduces them in LDS, one slot per needed pixel.
  //
  // Pass 1: occupancy bits of the tile's pixels, and per listed chunk which of its living points lie inside the tile.
  __device__ __forceinline__ void gather_bits() {
    constexpr int kU = 8;
    const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
    const int nitems = nlist << 6;
    const int lane = tid & 63;
    for (int e00 = 0; e00 < nitems; e00 += kU * NT) {
      int idx[kU];
      uint32_t p[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int e = e00 + u * NT + tid;
        idx[u] = -1;
        if (e < nitems && ((l_alive(e >> 6) >> (e & 63)) & 1ull)) idx[u] = (int)(l_chunk(e >> 6) << 6) + (e & 63);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) p[u] = idx[u] >= 0 && CHK(idx[u] < n_base && idx[u] < b.cap, 0) ? pixs[idx[u]] : 0u;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int e = e00 + u * NT + tid;
        int dl = -1, lp = -1;
        if (idx[u] >= 0) place_rc(p[u], dl, lp);
        if (POOL && planes_pooled) or_bits_by_runs(D, dl >= 0 ? lp : -1);
        else if (dl >= 0) D.set_local(lp);
        const unsigned long long msk = __ballot(dl >= 0);
        if (lane == 0 && e < nitems) set_kill(e >> 6, msk);    // (wave-uniform: the listed items come in 64s)
      }
    }
  }
  // Pass 2: the in-tile points (the masks pass 1 left) whose pixel the evaluation reads, min-reduced on the squared depth.
  __device__ __forceinline__ void gather_needed() {
    constexpr int kU = 4;
    const int n_head = uni(b.n_head[s]), n_virt = uni(w.n_virt[s]);
    const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
    const float4 *xyzi = reinterpret_cast<const float4 *>(b.xyzi) + (int64_t)s * b.cap;
    const int nitems = nlist << 6;
    for (int e00 = 0; e00 < nitems; e00 += kU * NT) {
      int idx[kU], slot[kU];
      uint32_t p[kU];
      float4 f[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int e = e00 + u * NT + tid;
        idx[u] = -1;
        if (e < nitems && ((l_kill(e >> 6) >> (e & 63)) & 1ull)) idx[u] = (int)(l_chunk(e >> 6) << 6) + (e & 63);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) p[u] = idx[u] >= 0 ? pixs[idx[u]] : 0u;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        slot[u] = -1;
        if (idx[u] < 0) continue;
        int dl, lp;
        place_rc(p[u], dl, lp);
        if (dl < 0) continue;
        const uint32_t nw = N.w[lp >> 5], bit = 1u << (lp & 31);
        if (nw & bit) slot[u] = (int)s_nrank[lp >> 5] + __popc(nw & (bit - 1u));
      }
      if (n_virt)                                              // (virtual order: the point of the slabs behind the listed number)
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (slot[u] >= 0) idx[u] = orig_of(w, b, s, n_virt, idx[u]);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        f[u] = make_float4(1.f, 0.f, 0.f, 0.f);
        if (slot[u] >= 0 && idx[u] < n_head) f[u] = xyzi[idx[u]];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (slot[u] < 0 || !CHK(slot[u] < nneed, 1)) continue;
        double x = (double)f[u].x, y = (double)f[u].y, z = (double)f[u].z;
        if (idx[u] >= n_head) load_point(b, s, idx[u], n_head, x, y, z);   // an inserted point: float64, from the log
        atomicMin(&s_ctile[slot[u]], depth_key(x * x + y * y + z * z));
      }
    }
  }
  // The scene side of an evaluation on a sparse tile, from the chunk list to the visible bits (steps 7-9 of scene_phase):
  // kOk; kNoFit when the room between `carve` and `list_start` does not hold it (nothing is lost: the caller goes on with
  // the dense routes); kStale.  On kOk `band_end` is where the tile ends.
  template <class Stale>
  __device__ __forceinline__ int scene_sparse(int carve, int list_start, bool serial, int &band_end, Stale &&stale) {
    const int cand_bytes = (ncand * 4 + 7) & ~7, img_bytes = (2 * ww * 4 + 7) & ~7;
    if ((int64_t)carve + cand_bytes + img_bytes + 256 > list_start) return kNoFit;
    g_dtile = nullptr;
    g_cand = nullptr;
    s_cand = reinterpret_cast<uint32_t *>(smem + carve);
    N.w = reinterpret_cast<uint32_t *>(smem + carve + cand_bytes);
    s_nrank = N.w + ww;
    const int tile_at = carve + cand_bytes + img_bytes;
    bt = dt;                                                   // (one band: the whole window)
    if (tid == 0) H[H_FILL] = H[H_CARRY] = 0;
    for (int e = tid; e < ww; e += NT) D.w[e] = 0u;
    __syncthreads();
    for (int e = tid; e < ww; e += NT) {                       // the candidates: where the sample is closed
      uint32_t bits = Cs.w[e];
      if (!bits) continue;
      int pos = atomicAdd(&H[H_FILL], __popc(bits));
      while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1;
        s_cand[pos++] = (uint32_t)((e << 5) + bit);
      }
    }
    STAMP(26);
    gather_bits();
    __syncthreads();
    const int nc = uni(H[H_FILL]);
    if (!serial && stale()) return kStale;
    closing(D, T, E);
    // holes of the scene among the candidates -> T; what the evaluation reads -> N
    for (int e = tid; e < ww; e += NT) T.w[e] = Cs.w[e] & ~D.w[e] & E.w[e];
    __syncthreads();
    morph(T.w, N.w, 0);
    for (int e = tid; e < ww; e += NT) N.w[e] = (N.w[e] | Cs.w[e]) & D.w[e];
    __syncthreads();
    for (int base = 0; base < ww; base += NT) {                // needed pixels before every window word
      const int e = base + tid;
      const int c = e < ww ? __popc(N.w[e]) : 0;
      int tot;
      const int ex = block_escan_i32(c, scan, tot);
      const int carry0 = H[H_CARRY];
      if (e < ww) s_nrank[e] = (uint32_t)(carry0 + ex);
      __syncthreads();
      if (tid == 0) H[H_CARRY] = carry0 + tot;
      __syncthreads();
    }
    nneed = uni(H[H_CARRY]);
    __syncthreads();                                           // (every wave has read the carry: see sample_phase)
    if ((int64_t)tile_at + (int64_t)nneed * 8 > list_start) {
      if (tid == 0) atomicAdd(&w.dbg[D_SPARSE_NOFIT], 1);
      return kNoFit;
    }
    band_end = tile_at + nneed * 8;
    s_ctile = reinterpret_cast<unsigned long long *>(smem + tile_at);
    for (int i = tid; i < nneed; i += NT) s_ctile[i] = R3D_SENT;
    for (int e = tid; e < ww; e += NT) T.w[e] = 0u;            // from here on: the visible pixels
    __syncthreads();
    gather_needed();
    __syncthreads();
    STAMP(27);                                                 // (both passes over the listed chunks, the scene's closing between them)
    for (int i = tid; i < nneed; i += NT) {                    // minima of the squared depth -> depths
      const unsigned long long k2 = s_ctile[i];
      if (k2 != R3D_SENT) s_ctile[i] = depth_key(sqrt(key_depth(k2)));
    }
    __syncthreads();
    STAMP(8);
    STAMP(9);
    if (tid == 0) atomicAdd(&w.dbg[D_SPARSE], 1);
    evaluate<true>(nc);
    __syncthreads();
    return kOk;
  }

  // ================================================================================================
  // scene phase against the first n_base_ points of the cloud.  kOk (results in LDS: visible bits
  // in T, s_V, kill masks in the chunk list, nvis, accept, header), kNoFit when this kernel's LDS
  // cannot hold the chunk list plus one band of the tile, or kNeedSerial when the scene has pixels
  // beyond 500 m and `serial` is false (their culling is not a matter of the window).
  // ================================================================================================
  // `stale()`: called by the whole workgroup at two points of a speculative evaluation (after the chunk list, after
  // the gather); true = a slot that finished meanwhile changed a pixel this evaluation reads, give up now (kStale).
  template <class Stale>
  __device__ __forceinline__ int scene_phase(int n_base_, bool serial, Stale &&stale) {
    n_base = uni(n_base_);
#ifdef R3D_CHECK
    if (!CHK(n_base >= 0 && n_base <= b.cap && n_base >= uni(b.n_head[s]), 0)) n_base = uni(b.n_head[s]);
#endif
    n_far = uni(b.n_far[s] < R3D_FAR_CAP ? b.n_far[s] : R3D_FAR_CAP);
    nvis = 0;
    accept = false;
    if (nvalid == 0) return kOk;
    RECORD_CHECK(1);                                           // on entry
    if (n_far > 0 && !serial) return kNeedSerial;
    // carve the scratch region: visible list | band: candidates, depth tile | hits | kill list | chunk list (from the end)
    int carve = r1;
    s_V = reinterpret_cast<uint16_t *>(smem + carve);
    carve = (carve + nvalid * 2 + 7) & ~7;
    const int W = dt.W;
    const int lds_end = lds_cap & ~15;
    const int min_band = 5 * W * 8 + W * 4;                   // one candidate row: 5 tile rows, W candidates
    // the chunk list in LDS, behind room for at least one band; in global memory when that leaves fewer than 64
    // entries (or when it overflows, below): k_insert_big in the scene's area, a chain pair in a piece of the pool
    s_list = smem + lds_end;
    if (lds_end - carve - min_band < 0) return kNoFit;
    const int n_chunks = (n_base + 63) >> 6;
    list_cap = (lds_end - carve - min_band) / kEntry;
    glist = force_glist || list_cap < 64;
    if (glist) {
      if (!force_glist) {
        const long long off = pool_take((long long)n_chunks * kEntry);
        if (off < 0) return kNoFit;
        g_list = w.tile_pool + off + (long long)n_chunks * kEntry;
      }
      list_cap = n_chunks;
    }

    if (tid == 0) {
      H[H_NLIST] = 0;
      H[H_CARRY] = 0;
      H[H_NVIS] = 0;
      H[H_REBASE] = 0;
      H[H_VRMIN] = H[H_VCMIN0] = H[H_VCMIN1] = 0x7FFFFFFF;
      H[H_VRMAX] = H[H_VCMAX0] = H[H_VCMAX1] = -1;
    }
    __syncthreads();

    STAMP(6);
    // -- 6. the chunks that can hold a point of the window --------------------------------------------
    // large clouds: first the super-boxes that reach the window's rows (their numbers in the room a band is given
    // later, in front of the list), then the boxes of their chunks only
    uint16_t *s_sup = nullptr;
    int nsup = 0;
    {
      const int n_sup = (n_chunks + 63) >> 6;
      const int room = glist ? lds_end - carve : min_band;
      if (supers_on(b, chunks) && n_sup <= 0xFFFF && n_sup * 2 <= room) {
        s_sup = reinterpret_cast<uint16_t *>(smem + carve);
        list_supers(s_sup, n_chunks);
        __syncthreads();
        nsup = uni(H[H_CARRY]);
      }
    }
    build_list(w.chunk_box + (int64_t)s * chunks, w.alive + (int64_t)s * chunks, n_chunks, s_sup, nsup);
    __syncthreads();
    nlist = uni(H[H_NLIST]);
    if (nlist > list_cap) {                                   // does not fit the LDS: once more, into global memory
      // (a racing predecessor's append can grow the box of the one partly filled chunk into the window between the
      // two passes: room for a few more than the first pass counted)
      if (!force_glist) {
        const long long off = pool_take((long long)(nlist + 8) * kEntry);
        if (off < 0) return kNoFit;
        g_list = w.tile_pool + off + (long long)(nlist + 8) * kEntry;
      }
      __syncthreads();
      if (tid == 0) H[H_NLIST] = 0;
      glist = true;
      list_cap = force_glist ? n_chunks : nlist + 8;
      __syncthreads();
      build_list(w.chunk_box + (int64_t)s * chunks, w.alive + (int64_t)s * chunks, n_chunks, s_sup, nsup);
      __syncthreads();
      // (alive bits only ever clear: a second pass lists at most the chunks of the first)
      nlist = uni(H[H_NLIST]) < list_cap ? uni(H[H_NLIST]) : list_cap;
    }
    RECORD_CHECK(2);                                           // chunk list built
    const int list_start = lds_end - (glist ? 0 : kEntry * nlist);
    const int band_bytes = list_start - carve;
    s_cand = reinterpret_cast<uint32_t *>(smem + carve);

    // rows of the candidates (the closed sample lies within 2 rows of a sample pixel); the tile as ONE
    // band when it fits (rows of the whole window: the occupancy bits then come from the tile), else
    // in bands of `per` candidate rows with 2 rows of halo on either side
    const int cr0 = uni(H[H_RMIN]) - 2 < 0 ? 0 : uni(H[H_RMIN]) - 2;
    const int cr1 = uni(H[H_RMAX]) + 2 > rows - 1 ? rows - 1 : uni(H[H_RMAX]) + 2;
    bool single = !(b.reserved & (kDbgBands | kDbgPoolTile)) && (int64_t)ncand * 4 + (int64_t)dt.npx * 8 + 8 <= band_bytes;
    // a window that does not fit the LDS: a sparse tile (round 6: only the pixels the evaluation reads, in LDS); when even
    // that does not fit, its tile and candidate list in a piece of the global pool (one band, the evaluation reads the
    // tile through L2); in row bands in LDS only when the pool is exhausted
    g_dtile = nullptr;
    g_cand = nullptr;
    sparse = false;
    int sparse_end = 0;
    // (only the shape for large range images carries the sparse tile: on the reference's grid a window beyond the 80 KB of a
    // 512-thread workgroup is rare -- 31 of 1 280 pairs of config C2 -- and the extra code cost that kernel 4 % of its launch)
    if (NT == 1024 && ((b.reserved & kDbgSparse) || (!single && !(b.reserved & (kDbgBands | kDbgPoolTile | kDbgNoSparse))))) {
      if (!serial && stale()) return kStale;
      STAMP(7);
      const int rs = scene_sparse(carve, list_start, serial, sparse_end, stale);
      if (rs == kStale) return kStale;
      sparse = rs == kOk;
      if (sparse) single = true;
    }
    // (round 5, measured on config C5: such tiles in row bands in LDS instead -- 30.1 against 5.07 ms per launch)
    if (!sparse && !single && !(b.reserved & (kDbgBands | kDbgNoPoolTile)) && pool_off != -2) {
      if (pool_off == -1) {
        pool_off = pool_take((((long long)dt.npx * 8 + 255) & ~255ll) + (long long)ncand * 4);
        if (pool_off < 0) pool_off = -2;
      }
      if (pool_off >= 0) {
        if (tid == 0) atomicAdd(&w.dbg[D_TILE_POOLED], 1);
        g_dtile = reinterpret_cast<unsigned long long *>(w.tile_pool + pool_off);
        g_cand = reinterpret_cast<uint32_t *>(w.tile_pool + pool_off + (((long long)dt.npx * 8 + 255) & ~255ll));
        single = true;
      }
    }
    lazy_root = NT == 1024 && g_dtile != nullptr && dt.npx > 8 * ncand;
    int per = cr1 - cr0 + 1;
    if (!single) {
      per = (band_bytes - 4 * W * 8 - 8) / (12 * W);
      if (b.reserved & kDbgBands) per = per > 3 ? 3 : per;
      if (per < 1) return kNoFit;
    }
    // one band: behind it (behind `carve` when the tile is pooled) the numbers of the entries the kill pass looks at
    // (from the chunk list downwards) and the hits, up to the chunk list
    intile = single;
    s_hit = nullptr;
    hit_cap = 0;
    uint16_t *s_kl = nullptr;
    int kl_room = 0;
    if (single) {
      const int band_end = sparse ? sparse_end : (g_dtile ? carve : (((carve + ncand * 4 + 7) & ~7) + dt.npx * 8));
      const int kl_bytes = nlist <= 0xFFFF ? (nlist * 2 + 7) & ~7 : 0;
      if (band_end + kl_bytes <= list_start && kl_bytes) {
        s_kl = reinterpret_cast<uint16_t *>(smem + list_start - kl_bytes);
        kl_room = nlist;
      }
      const int hit_bytes = list_start - (s_kl ? kl_bytes : 0) - band_end;
      // (tile and window pixel of a hit share a word: fewer than 65 536 of either)
      if (!sparse && NT <= 512 && hit_bytes >= 512 && (ww << 5) < 0xFFFF && dt.npx < 0xFFFF && !(b.reserved & kDbgNoHits)) {
        s_hit = reinterpret_cast<uint2 *>(smem + band_end);
        hit_cap = hit_bytes >> 3;
      }
    }
    flat = !sparse && NT <= 512 && single && (ww << 5) < 0xFFFF && dt.npx < 0xFFFF;
    if (tid == 0) {
      H[H_NHIT] = 0;
      H[H_HITEND] = 0x7FFFFFFF;
    }

    if (!sparse && !serial && stale()) return kStale;
    if (!sparse) STAMP(7);
    WinImage &vis = T;
    for (int a0 = cr0; a0 <= cr1 && !sparse; a0 += per) {
      const int a1 = a0 + per - 1 > cr1 ? cr1 : a0 + per - 1;
      // -- 7. this band of the scene's range image, from the living points -----------------------------
      bt = dt;
      if (!single) {
        bt.r0 = a0 - 2 < dt.r0 ? dt.r0 : a0 - 2;
        bt.r1 = a1 + 2 > dt.r1 ? dt.r1 : a1 + 2;
      }
      bt.npx = (bt.r1 - bt.r0 + 1) * W;
      // candidates of the band's rows first (their count decides where the tile starts); with bands,
      // also the listed chunks whose rows reach this band (a scan in ring order has one-row chunks)
      if (tid == 0) H[H_FILL] = H[H_CARRY] = 0;
      __syncthreads();
      for (int e = (a0 - win.r_lo) * win.njw + tid; e < (a1 - win.r_lo + 1) * win.njw; e += NT) {
        uint32_t bits = Cs.w[e];
        if (!bits) continue;
        int pos = atomicAdd(&H[H_FILL], __popc(bits));
        while (bits) {
          int bit = __ffs(bits) - 1;
          bits &= bits - 1;
          if (g_cand) g_cand[pos++] = (uint32_t)((e << 5) + bit);
          else s_cand[pos++] = (uint32_t)((e << 5) + bit);
        }
      }
      const bool first = a0 == cr0;
      uint16_t *s_sub = s_V;                                   // the visible list is built after the last band
      const bool use_sub = !single && !first && nlist <= nvalid;
      if (use_sub)
        for (int i = tid; i < nlist; i += NT) {
          uint32_t rr = l_rows(i);
          if ((int)(rr & 0xFFFF) <= bt.r1 && (int)(rr >> 16) >= bt.r0) s_sub[atomicAdd(&H[H_CARRY], 1)] = (uint16_t)i;
        }
      __syncthreads();
      const int nc = uni(H[H_FILL]), nsub = uni(H[H_CARRY]);
      s_dtile = reinterpret_cast<unsigned long long *>(smem + ((carve + nc * 4 + 7) & ~7));
      if (g_dtile)
        for (int i = tid; i < bt.npx; i += NT) g_dtile[i] = R3D_SENT;
      else
        for (int i = tid; i < bt.npx; i += NT) s_dtile[i] = R3D_SENT;
      // the scene's occupancy bits: set by the gather when the tile is banded or lives in the pool (reading a
      // pooled tile back costs a trip through L2 per pixel), else read off the finished tile in LDS below
      const bool bits_in_gather = first && (!single || g_dtile);
      if (bits_in_gather)
        for (int e = tid; e < ww; e += NT) D.w[e] = 0u;
      __syncthreads();
      if (first) STAMP(26);                                    // (candidates listed, tile and occupancy bits cleared)
      if (flat) gather_flat(bits_in_gather);
      else gather(bits_in_gather, use_sub ? s_sub : nullptr, nsub);
      if (first) STAMP(27);                                    // (gathered)
      RECORD_CHECK(3);
      if (g_dtile) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the minima were formed in L2: drop this CU's copies
      __syncthreads();
      // a band in LDS holds minima of the squared depth: the root of every occupied pixel (and, for one band, the
      // scene's occupancy bits off the same pass)
      if (!lazy_root) {                                        // (else: the root per read, tile_key)
        const bool bits_here = first && single && !g_dtile;
        if (bits_here)
          for (int e = tid; e < ww; e += NT) D.w[e] = 0u;
        if (bits_here) __syncthreads();
        for (int i = tid; i < bt.npx; i += NT) {
          const unsigned long long k2 = g_dtile ? g_dtile[i] : s_dtile[i];
          if (k2 == R3D_SENT) continue;
          const unsigned long long k1 = depth_key(sqrt(key_depth(k2)));
          if (g_dtile) g_dtile[i] = k1;
          else s_dtile[i] = k1;
          if (bits_here) {
            int k, r = bt.r0 + by_W.div(i, k);
            int c = k < dt.w0 ? dt.c00 + k : dt.c01 + (k - dt.w0);
            D.set_local(win.lpix_rc(r, c));
          }
        }
        __syncthreads();
      }

      if (first) {
        if (!serial && stale()) return kStale;
        STAMP(8);
        // -- 8. closing of the scene's occupancy ------------------------------------------------------
        RECORD_CHECK(4);                                       // tile roots and scene bits
        closing(D, T, E);
        for (int e = tid; e < ww; e += NT) T.w[e] = 0u;        // from here on: the visible pixels
        __syncthreads();
        STAMP(9);
        RECORD_CHECK(5);                                       // scene closed
      }

      // -- 9. visibility on the candidates: smoothed sample depth < smoothed scene depth (:461-467) ----
      evaluate<false>(nc);
      __syncthreads();
    }

    STAMP(10);
    RECORD_CHECK(6);                                           // evaluated
    // -- 10. accept test (insertion.py:511-517); the visible points in order; who dies -----------------
    nvis = uni(H[H_NVIS]);
    // cull_only (the state a REJECTED candidate leaves, see commit): the reference culls the scene in EVERY visible pixel,
    // also in one that holds no sample point -- a closing-filled hole of the sample in front of the scene (insertion.py:467-473
    // runs before len(visible_sample) is looked at, :511) -- so a candidate rejected with no visible point can still leave a
    // culled copy: any visible pixel decides, not the count
    accept = cull_only ? uni(H[H_VRMAX]) >= 0 : (nvis > 0 && nvis >= need);
    if (accept) {
      // the visible points in sorted order: thread t takes the sorted points [t*L, t*L + L), one block scan
      {
        const int L = (nvalid + NT - 1) / NT;
        const int k_lo = tid * L < nvalid ? tid * L : nvalid, k_hi = k_lo + L < nvalid ? k_lo + L : nvalid;
        int cnt = 0;
        for (int k = k_lo; k < k_hi; ++k) cnt += vis.get_local((int)s_lp[s_F[k]]) ? 1 : 0;
        int tot;
        int o = block_escan_i32(cnt, scan, tot);
#ifdef R3D_CHECK
        if (tid == 0 && tot != nvis) atomicAdd(&w.dbg[15], 1);        // the visible list is not as long as the count says
#endif
        for (int k = k_lo; k < k_hi; ++k)
          if (vis.get_local((int)s_lp[s_F[k]])) s_V[o++] = (uint16_t)k;
      }
      STAMP(31);                                               // (visible list made)
      // every living scene point in a visible pixel dies (:470-473): one mask per listed chunk, so that
      // the commit is a handful of atomics.  Only the chunks whose rows reach a visible row are looked at (and, with
      // one band, only those with a point inside the tile): their entry numbers are compacted first.  A chunk whose
      // in-tile points were kept as hits needs no second trip to its pixel ids.
      {
        constexpr int kPer = NT == 1024 ? 8 : 4;             // (the 1024-thread shape keeps no hits: its pixel ids come from memory)
        const int lane = tid & 63;
        const uint32_t *pixs = reinterpret_cast<const uint32_t *>(b.pix) + (int64_t)s * b.cap;
        const int vr0 = uni(H[H_VRMIN]), vr1 = uni(H[H_VRMAX]);
        if (!single) {                                         // bands: the room the candidates and the tile no longer need
          kl_room = (list_start - carve) / 2;
          s_kl = reinterpret_cast<uint16_t *>(smem + carve);
          if (nlist > kl_room || nlist > 0xFFFF) s_kl = nullptr;
        }
        int nkl = nlist;
        if (s_kl) {
          if (tid == 0) H[H_CARRY] = 0;
          __syncthreads();
          for (int i = tid; i < nlist; i += NT) {
            uint32_t rr = l_rows(i);
            const bool reach = (int)(rr & 0xFFFF) <= vr1 && (int)(rr >> 16) >= vr0 && (!intile || l_kill(i) != 0ull);
            if (reach) s_kl[atomicAdd(&H[H_CARRY], 1)] = (uint16_t)i;
            else if (intile) set_kill(i, 0ull);                // (the field held the in-tile mask)
          }
          __syncthreads();
          nkl = uni(H[H_CARRY]);
        }
        const int nitems = nkl << 6;
        for (int e0 = tid; e0 < nitems; e0 += kPer * NT) {
          int lpv[kPer], ent[kPer];
          uint32_t pg[kPer];
          bool on[kPer], from_hits[kPer];
#pragma unroll
          for (int u = 0; u < kPer; ++u) {
            int e = e0 + u * NT;
            ent[u] = e < nitems ? (s_kl ? (int)s_kl[e >> 6] : (e >> 6)) : 0;
            // the living points of the chunk -- those inside the tile, when the gather has left their mask
            const unsigned long long msk = e < nitems ? (intile ? l_kill(ent[u]) : l_alive(ent[u])) : 0ull;
            on[u] = (msk >> (e & 63)) & 1ull;
            const int hb = flat && e < nitems ? l_hbase(ent[u]) : -1;
            from_hits[u] = hb >= 0;
            lpv[u] = -1;
            pg[u] = 0u;
            if (on[u]) {
              if (from_hits[u]) {
                const int at = hb + __popcll(msk & ((1ull << (e & 63)) - 1ull));
                lpv[u] = CHK(at < hit_cap, 4) ? (int)(s_hit[at].y & 0xFFFFu) : -1;
              } else if (CHK((int)l_chunk(ent[u]) < chunks, 5))
                pg[u] = pixs[(int)(l_chunk(ent[u]) << 6) + (e & 63)];
            }
          }
#pragma unroll
          for (int u = 0; u < kPer; ++u) {
            int e = e0 + u * NT;
            bool kill = false;
            if (on[u]) {
              int lp = lpv[u], dl_unused;
              if (!from_hits[u]) place_rc(pg[u], dl_unused, lp);
              kill = lp >= 0 && vis.get_local(lp);
            }
            unsigned long long mask = __ballot(kill);
            if (lane == 0 && e < nitems && (mask || intile)) set_kill(ent[u], mask);
          }
        }
        // A point that holds an elevation bound (z/r bit-equal to the extreme, w.q_ext) sits in the first or the last
        // row of the image; if it dies the bounds may move and the scene is re-based (insertion.py:373 recomputes them
        // from the merged cloud for every insert; a rebase that was not needed changes nothing).  Only a pair whose
        // visible pixels reach one of those rows has to look: at the points it culls.
        if (vr0 == 0 || vr1 == rows - 1 || tiny_el) {
          __syncthreads();
          const int n_head = uni(b.n_head[s]);
          const double q_min = w.q_ext[2 * s + 0], q_max = w.q_ext[2 * s + 1];
          for (int e = tid; e < (nlist << 6); e += NT) {
            const int i = e >> 6;
            if (!((l_kill(i) >> (e & 63)) & 1ull)) continue;
            double x, y, z;
            load_point(b, s, orig_of(w, b, s, uni(w.n_virt[s]), (int)(l_chunk(i) << 6) + (e & 63)), n_head, x, y, z);
            const double q = z / sqrt(x * x + y * y + z * z);
            if (q == q_min || q == q_max) atomicOr(&H[H_REBASE], 2);
          }
        }
      }
      __syncthreads();
    }
    STAMP(11);
    RECORD_CHECK(7);                                           // visible list and kill masks
    return kOk;
  }

  // ================================================================================================
  // commit of an accepted candidate: append (insertion.py:526), cull (:470-473).  Returns true when
  // the scene must be re-based (the elevation bounds may have moved).  `flags_out`: kRec* bits.
  // ================================================================================================
  // What a commit is made from: the evaluation's structures in this workgroup's LDS (FromLds), or the record a parked pair
  // has left in the pool (FromRecord; written by park_write() through the very accessors of FromLds).
  struct FromLds {
    static constexpr bool kLds = true;
    const Ins &I;
    __device__ __forceinline__ void vis(int o, int &j, int &row, int &col) const {     // visible point o, in (pixel, index) order
      j = I.s_F[I.s_V[o]];
      const int lp = (int)I.s_lp[j];
      I.win.row_word(lp >> 5, row, col);
      col = (col << 5) + (lp & 31);
    }
    __device__ __forceinline__ int n_kill() const { return I.nlist; }
    __device__ __forceinline__ bool kill(int i, int &c, unsigned long long &mask) const {   // false: the chunk loses nobody
      mask = I.l_kill(i);
      c = (int)I.l_chunk(i);
      return mask != 0ull;
    }
  };
  struct FromRecord {
    static constexpr bool kLds = false;
    const uint2 *v;                 // {sample index, packed pixel} per visible point
    const ulonglong2 *kl;           // {chunk, mask}
    int nkill;
    __device__ __forceinline__ void vis(int o, int &j, int &row, int &col) const {
      const uint2 e = v[o];
      j = (int)e.x;
      row = pix_row(e.y);
      col = pix_col(e.y);
    }
    __device__ __forceinline__ int n_kill() const { return nkill; }
    __device__ __forceinline__ bool kill(int i, int &c, unsigned long long &mask) const {
      const ulonglong2 e = kl[i];
      c = (int)e.x;
      mask = e.y;
      return mask != 0ull;
    }
  };
  static __device__ __forceinline__ long long park_vis_bytes(int nvis_) { return ((long long)nvis_ * 8 + 15) & ~15ll; }

  // Leaves what commit() needs of this (accepted) evaluation in a piece of the launch's pool.  Returns its offset, -1 when
  // the pool is exhausted; `n_kill_out`: entries of the kill list.  Whole workgroup.
  __device__ __forceinline__ long long park_write(int &n_kill_out) {
    n_kill_out = 0;
    if (!accept) return 0;
    const long long off = pool_take(park_vis_bytes(nvis) + (long long)nlist * 16);
    if (off < 0) return -1;
    uint2 *v = reinterpret_cast<uint2 *>(w.tile_pool + off);
    ulonglong2 *kl = reinterpret_cast<ulonglong2 *>(w.tile_pool + off + park_vis_bytes(nvis));
    const FromLds src{*this};
    for (int o = tid; o < nvis; o += NT) {
      int j, row, col;
      src.vis(o, j, row, col);
      v[o] = make_uint2((uint32_t)j, pack_pix(row, col));
    }
    if (tid == 0) H[H_CARRY] = 0;
    __syncthreads();
    for (int i = tid; i < nlist; i += NT) {
      int c;
      unsigned long long mask;
      if (src.kill(i, c, mask)) kl[atomicAdd(&H[H_CARRY], 1)] = make_ulonglong2((unsigned long long)(uint32_t)c, mask);
    }
    __syncthreads();
    n_kill_out = uni(H[H_CARRY]);
    return off;
  }
  __device__ __forceinline__ bool sample_far() const { return uni(H[H_SFAR]) != 0; }

  // n_total_in / n_head_in: the scene's counts when the caller already holds them (the chain kernel: from the
  // predecessor's record), -1 = read them here.  The log holds one row per appended point: n_log = n_total - n_head.
  template <class SRC>
  __device__ __forceinline__ bool commit(const SRC &src, int &flags_out, int &n_total_after, int n_total_in = -1, int n_head_in = -1) {
    const int lane = tid & 63, wave = tid >> 6;
    const int n_head = n_head_in >= 0 ? n_head_in : uni(b.n_head[s]);
    const int n_total = n_total_in >= 0 ? n_total_in : uni(b.n_total[s]);
    const int n_log = n_total - n_head;
    const int tiles = (int)((b.cap + kTile - 1) / kTile);
    n_total_after = n_total;
    flags_out = 0;
    if (accept && !cull_only && ((int64_t)n_total + nvis > b.cap || (int64_t)n_log + nvis > b.log_cap)) {
      accept = false;
      if (tid == 0) atomicOr(&b.status[s], R3D_S_CAPACITY);
    }
    if (tid == 0 && H[H_FLAGS]) atomicOr(&b.status[s], H[H_FLAGS]);
    unsigned long long *alive = w.alive + (int64_t)s * chunks;
    int32_t *tile_alive = w.tile_alive + (int64_t)s * tiles;
    // min_points < 0: what the reference's driver is left with after a REJECTED candidate -- the scene without the points
    // the candidate covers and without the candidate (insertion.py:468-471 without :526; the copy stays bound to scene_pcl
    // until the next candidate restores the backup, :453).  It goes into a SHADOW of the alive bits: the scene itself is
    // untouched, r3d_batch_export_rows shows the copy, r3d_batch_adopt_rejected makes it the scene.
    if (cull_only) {
      unsigned long long *sh = w.alive_shadow + (int64_t)s * chunks;
      int32_t *ts = w.tile_shadow + (int64_t)s * tiles;
      const int n_chunks = (n_total + 63) >> 6;
      for (int c = tid; c < n_chunks; c += NT) sh[c] = alive[c];
      for (int t = tid; t < tiles; t += NT) ts[t] = tile_alive[t];
      phase_sync();
      if (accept) {
        for (int i = tid; i < src.n_kill(); i += NT) {
          int c;
          unsigned long long mask;
          if (!src.kill(i, c, mask)) continue;
          if (!CHK(c < chunks, 5)) continue;
          atomicAnd(&sh[c], ~mask);
          atomicSub(&ts[(c << 6) / kTile], __popcll(mask));
        }
        if constexpr (SRC::kLds)
          if (n_far > 0) far_pass(n_total, sh, ts);
      }
      __syncthreads();
      if (tid == 0) w.shadow_valid[s] = accept ? 1 : 0;
      accept = false;
      return false;
    }
    if (tid == 0) w.shadow_valid[s] = 0;                      // an evaluated candidate starts from the backup (:453)
    if (!accept) return false;
    STAMP(22);
    // -- the visible points, in (pixel, index) order, behind the cloud; one 64-point chunk per wave step
    {
      const int c_first = n_total >> 6, c_last = (n_total + nvis - 1) >> 6;
      for (int ci = c_first + wave; ci <= c_last; ci += NT / 64) {
        int dst = (ci << 6) + lane, o = dst - n_total;
        bool valid = o >= 0 && o < nvis;
        BoxAcc box;
#ifdef R3D_CHECK
        if constexpr (SRC::kLds)
          if (valid) {                                          // which of the three: room | the visible list | the sorted order
            const bool c_room = dst < b.cap && n_log + o < b.log_cap && dst >= n_head;
            const bool c_v = (int)s_V[o] < nvalid;
            const bool c_f = c_v && (int)s_F[s_V[o]] < m;
            if (!c_room) atomicAdd(&w.dbg[14], 1);
            if (!c_v || !c_f) {
              atomicAdd(&w.dbg[14], 1);
            }
            if (!c_room || !c_v || !c_f) valid = false;
          }
#endif
        if (valid) {
          int j, row, col;
          src.vis(o, j, row, col);
          const double *q = rows5 + (int64_t)j * 5;
          double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
          int lr = n_log + o;
          float4 f;
          f.x = (float)q0;
          f.y = (float)q1;
          f.z = (float)q2;
          f.w = (float)q3;
          reinterpret_cast<float4 *>(b.xyzi)[(int64_t)s * b.cap + dst] = f;
          b.label[(int64_t)s * b.cap + dst] = (uint32_t)(int64_t)q4;
          b.pix[(int64_t)s * b.cap + dst] = (int32_t)pack_pix(row, col);
          b.tail_ref[(int64_t)s * b.log_cap + (dst - n_head)] = lr;
          double *l = b.log5 + ((int64_t)s * b.log_cap + lr) * 5;
          l[0] = q0;
          l[1] = q1;
          l[2] = q2;
          l[3] = q3;
          l[4] = q4;
          b.log_birth[(int64_t)s * b.log_cap + lr] = step;
          box.add(row, col);
        }
        const bool old_chunk = (ci << 6) < n_total;           // holds earlier points: extend its box
        if (lane == 0 && old_chunk) box.add_box(w.chunk_box[(int64_t)s * chunks + ci], cols);
        unsigned long long packed = box.wave_pack();
        unsigned long long living = __ballot(valid);
        if (lane == 0) {
          w.chunk_box[(int64_t)s * chunks + ci] = packed;
          if (supers_on(b, chunks) && (int)(packed & 0xFFFF) <= (int)((packed >> 16) & 0xFFFF)) {
            int32_t *sr = w.super_rows + ((int64_t)s * ((chunks + 63) >> 6) + (ci >> 6)) * 2;
            atomicMin(&sr[0], (int)(packed & 0xFFFF));
            atomicMax(&sr[1], (int)((packed >> 16) & 0xFFFF));
          }
          if (old_chunk) atomicOr(&alive[ci], living);
          else alive[ci] = living;
          atomicAdd(&tile_alive[(ci << 6) / kTile], __popcll(living));
        }
      }
    }
    STAMP(23);
    // -- the scene points in visible pixels die
    for (int i = tid; i < src.n_kill(); i += NT) {
      int c;
      unsigned long long mask;
      if (!src.kill(i, c, mask)) continue;
      if (!CHK(c < chunks, 5)) continue;
#ifdef R3D_CHECK
      {
        const unsigned long long was = atomicAnd(&alive[c], ~mask);
        CHK((was & mask) == mask, 6);                        // every culled point was alive
      }
#else
      atomicAnd(&alive[c], ~mask);
#endif
      atomicSub(&tile_alive[(c << 6) / kTile], __popcll(mask));
    }
    STAMP(24);
    if constexpr (SRC::kLds) {
      // -- pixels that now hold a return beyond 500 m join the far list (a sample with such a point is never committed
      // from a record: sample_far())
      if (uni(H[H_SFAR]))
        for (int o = tid; o < nvis; o += NT) {
          int k = s_V[o];
          int lp = (int)s_lp[s_F[k]];
          int rk = rank_of(lp);
          if (k != (int)s_start[rk]) continue;                    // once per pixel
          if (key_depth(s_sdepth[rk]) > R3D_EMPTY_DEPTH) {
            int f = atomicAdd(&b.n_far[s], 1);
            if (f < R3D_FAR_CAP) b.far_pix[(int64_t)s * R3D_FAR_CAP + f] = global_pix(lp);
            else atomicOr(&b.status[s], R3D_S_FAR_OVERFLOW);
            H[H_FARADD] = 1;
          }
        }
      // -- scene pixels deeper than 500 m are visible to any accepted insert whose sample is empty there
      // (500 < depth, insertion.py:99,:467): not a matter of the window.  Rare: two passes over the cloud.
      if (n_far > 0) far_pass(n_total);
    }
    __syncthreads();
    STAMP(25);
    const bool rebase = uni(H[H_REBASE]) != 0;
    flags_out = kRecAccepted | (rebase ? kRecRebased : 0) | ((H[H_FARADD] || n_far > 0) ? kRecFar : 0);
    n_total_after = n_total + nvis;
    if (tid == 0) {
      b.n_total[s] = n_total + nvis;
      b.n_log[s] = n_log + nvis;
      if (rebase) {
        b.rebase[s] += 1;                                     // single writer per scene
        // why: a visible sample point outside the bounds | a culled point held a bound | ... found by the far pass
        const int why = H[H_REBASE];
        if (why & 1) atomicAdd(&w.dbg[D_REBASE_OOB], 1);
        if (why & 2) atomicAdd(&w.dbg[D_REBASE_HOLDER], 1);
        if (why & 4) atomicAdd(&w.dbg[D_REBASE_FAR], 1);
        if (!(why & 7)) atomicAdd(&w.dbg[D_REBASE_OTHER], 1);
      }
    }
    return rebase;
  }

  // Diagnostic (bit 64): an order-independent digest of what an evaluation decided -- visible count, accept, rebase
  // flag, the visible pixels, and for an accepted pair which points of which chunk die.  Whole workgroup.
  __device__ __forceinline__ unsigned long long signature() {
    if (nvalid == 0) return 0ull;
    unsigned long long *cell = reinterpret_cast<unsigned long long *>(&H[H_SIG]);
    __syncthreads();
    if (tid == 0) *cell = 0ull;
    __syncthreads();
    unsigned long long h = 0ull;
    for (int e = tid; e < ww; e += NT) h += (unsigned long long)T.w[e] * (0x9E3779B97F4A7C15ull * (unsigned long long)(e + 1));
    if (accept)
      for (int i = tid; i < nlist; i += NT) {
        const unsigned long long mk = l_kill(i);
        if (mk) h += (mk ^ (mk >> 29)) * (0xC2B2AE3D27D4EB4Full * (unsigned long long)(l_chunk(i) + 1u));
      }
    if (h) atomicAdd(cell, h);
    __syncthreads();
    const unsigned long long sum = *cell;
    __syncthreads();
    return sum + (unsigned long long)nvis * 1000003ull + (accept ? 7ull : 0ull) + (H[H_REBASE] ? 13ull : 0ull);
  }

  // The far pixels that are not candidates of this insert: smoothed sample depth 500 there, the scene's
  // depth is the raw minimum (the pixel is occupied): visible iff that minimum exceeds 500.
  __device__ __forceinline__ void far_pass(int n_total, unsigned long long *kill_alive = nullptr, int32_t *kill_tiles = nullptr) {
    const int lane = tid & 63;
    const int n_head = b.n_head[s];
    const int tiles = (int)((b.cap + kTile - 1) / kTile);
    uint32_t *fpx = w.cand + (int64_t)s * w.cand_stride;      // the scene's scratch: far pixels, their minima
    unsigned long long *fmin = reinterpret_cast<unsigned long long *>(fpx + R3D_FAR_CAP);
    const double q_min = w.q_ext[2 * s + 0], q_max = w.q_ext[2 * s + 1];
    const int32_t *pixs = b.pix + (int64_t)s * b.cap;
    unsigned long long *alive = w.alive + (int64_t)s * chunks;
    int32_t *tile_alive = w.tile_alive + (int64_t)s * tiles;
    if (!kill_alive) kill_alive = alive, kill_tiles = tile_alive;   // (else: the shadow of a rejected candidate)
    __syncthreads();
    for (int f = tid; f < n_far; f += NT) {
      uint32_t p = (uint32_t)b.far_pix[(int64_t)s * R3D_FAR_CAP + f];
      int lp = win.lpix_rc(pix_row(p), pix_col(p));
      bool is_cand = lp >= 0 && Cs.get_local(lp);
      fpx[f] = is_cand ? 0xFFFFFFFFu : p;
      fmin[f] = R3D_SENT;
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
      for (int i0 = 0; i0 < n_total; i0 += NT) {
        int i = i0 + tid;
        bool kill = false;
        if (i < n_total && ((__hip_atomic_load(&alive[i >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (i & 63)) & 1ull)) {
          uint32_t p = (uint32_t)pixs[i];
          int hit = -1;
          for (int f = 0; f < n_far; ++f)
            if (fpx[f] == p) {
              hit = f;
              break;
            }
          if (hit >= 0) {
            double x, y, z;
            load_point(b, s, orig_of(w, b, s, w.n_virt[s], i), n_head, x, y, z);
            double r = sqrt(x * x + y * y + z * z);
            if (pass == 0) atomicMin(&fmin[hit], depth_key(r));
            else if (key_depth(__hip_atomic_load(&fmin[hit], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > R3D_EMPTY_DEPTH) {
              kill = true;
              double q = z / r;
              if (q == q_min || q == q_max) atomicOr(&H[H_REBASE], 4);
            }
          }
        }
        if (pass == 1) {
          unsigned long long mask = __ballot(kill);
          if (lane == 0 && mask) {
            atomicAnd(&kill_alive[(i0 + tid) >> 6], ~mask);
            atomicSub(&kill_tiles[(i0 + tid) / kTile], __popcll(mask));
          }
        }
      }
      __syncthreads();
    }
  }
};


// The slot's inputs for scene s; false when the slot has nothing to evaluate there.
template <class INS>
__device__ __forceinline__ bool load_slot(INS &I, const r3d_batch_t &b, const ChainSlots &slots, int k, int s,
                                          int first_step) {
  const int64_t off = slots.sample_off[k][s];
  const int64_t m64 = slots.sample_off[k][s + 1] - off;
  const bool act = !slots.active[k] || slots.active[k][s];
  I.rows5 = slots.samples5[k] + off * 5;
  I.m = uni((int)m64);
  I.need = uni(slots.min_points[k][s]);
  I.cull_only = I.need < 0;
  I.step = first_step + k;
  if (act && m64 > kKeyCap && threadIdx.x == 0) atomicOr(&b.status[s], R3D_S_SAMPLE_TOO_LARGE);
  return act && m64 > 0 && m64 <= kKeyCap;
}

}  // namespace r3d
