// Host side of the file-to-file pipeline: the packer that lays the frames of
// a batch out in the pinned staging buffer the way r3d_batch_t wants them on the device.  Plain C++
// (std::thread), no HIP: the copies into HBM are the caller's (hipMemcpyAsync on its copy streams).
#include <cerrno>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <emmintrin.h>
#include <fcntl.h>
#include <limits.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include "r3d_host.hpp"

namespace {

// The slabs are written once and read next by somebody else (the copy engine, the file writers): stores that go
// around the caches.  An ordinary store first reads the line it is about to overwrite; at 2.5 MB per frame and
// 17 000 frames per second that read-for-ownership is a quarter of the host's memory traffic, and the host's
// memory system is what bounds the streamed pipeline (tools/e2e_stages.py).  SSE2 only: part of every x86-64.
inline void stream_copy(void *dst, const void *src, size_t bytes) {
  char *d = static_cast<char *>(dst);
  const char *s = static_cast<const char *>(src);
  size_t head = (size_t)(-(uintptr_t)d) & 63;
  if (head > bytes) head = bytes;
  std::memcpy(d, s, head);
  d += head, s += head, bytes -= head;
  for (; bytes >= 64; d += 64, s += 64, bytes -= 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 32));
    const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 48));
    _mm_stream_si128(reinterpret_cast<__m128i *>(d), a);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + 48), e);
  }
  std::memcpy(d, s, bytes);
}

// dst[i] = the label the pipeline keeps of src[i]: the semantic half (SS tools/datasets.py:53-55), or collapsed to
// {keep, 1} (OD insertion.py:353-355); dst == src is allowed
inline void stream_labels(uint32_t *dst, const uint32_t *src, int64_t n, int32_t collapse_keep) {
  int64_t i = 0;
  auto one = [&](uint32_t v) {
    v &= 0xFFFFu;
    return collapse_keep < 0 ? v : (v == (uint32_t)collapse_keep ? (uint32_t)collapse_keep : 1u);
  };
  for (; i < n && ((uintptr_t)(dst + i) & 63); ++i) dst[i] = one(src[i]);
  const __m128i mask = _mm_set1_epi32(0xFFFF), keep = _mm_set1_epi32(collapse_keep), ones = _mm_set1_epi32(1);
  for (; i + 16 <= n; i += 16)
    for (int j = 0; j < 16; j += 4) {
      __m128i v = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + j)), mask);
      if (collapse_keep >= 0) {
        const __m128i eq = _mm_cmpeq_epi32(v, keep);
        v = _mm_or_si128(_mm_and_si128(eq, keep), _mm_andnot_si128(eq, ones));
      }
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + j), v);
    }
  for (; i < n; ++i) dst[i] = one(src[i]);
}

// dst3[i] = x y z of src4[i] (rows of x y z intensity), 12 bytes per point: what the device needs of a frame in delta mode
// (r3d_batch_begin_xyz).  Four points per step: 64 bytes in, 48 out, streaming stores (dst3 16-byte aligned).
inline void stream_xyz(float *dst3, const float *src4, int64_t n) {
  int64_t i = 0;
  if (((uintptr_t)dst3 & 15) == 0)
    for (; i + 4 <= n; i += 4) {
      const __m128 a = _mm_loadu_ps(src4 + 4 * i), b = _mm_loadu_ps(src4 + 4 * i + 4), c = _mm_loadu_ps(src4 + 4 * i + 8),
                   d = _mm_loadu_ps(src4 + 4 * i + 12);
      const __m128 ab = _mm_shuffle_ps(a, b, _MM_SHUFFLE(0, 0, 2, 2));      // z0 z0 x1 x1
      const __m128 cd = _mm_shuffle_ps(c, d, _MM_SHUFFLE(0, 0, 2, 2));      // z2 z2 x3 x3
      _mm_stream_ps(dst3 + 3 * i, _mm_shuffle_ps(a, ab, _MM_SHUFFLE(2, 0, 1, 0)));        // x0 y0 z0 x1
      _mm_stream_ps(dst3 + 3 * i + 4, _mm_shuffle_ps(b, c, _MM_SHUFFLE(1, 0, 2, 1)));     // y1 z1 x2 y2
      _mm_stream_ps(dst3 + 3 * i + 8, _mm_shuffle_ps(cd, d, _MM_SHUFFLE(2, 1, 2, 0)));    // z2 x3 y3 z3
    }
  for (; i < n; ++i) {
    dst3[3 * i + 0] = src4[4 * i + 0];
    dst3[3 * i + 1] = src4[4 * i + 1];
    dst3[3 * i + 2] = src4[4 * i + 2];
  }
}

}  // namespace

extern "C" {

int r3d_host_pack_frames(const float *const *xyzi, const uint32_t *const *label, const int32_t *n_points, int32_t B,
                         int64_t cap, float *dst_xyzi, uint32_t *dst_label, int32_t collapse_keep, int32_t threads) {
  return r3d_host_pack_frames_xyz(xyzi, label, n_points, B, cap, dst_xyzi, dst_label, nullptr, collapse_keep, threads);
}

int r3d_host_read_frames(const char *const *velodyne_paths, const char *const *label_paths, int32_t B, int64_t cap, float *dst_xyzi,
                         uint32_t *dst_label, int32_t *n_points, int32_t collapse_keep, int32_t threads) {
  return r3d_host_read_frames_xyz(velodyne_paths, label_paths, B, cap, dst_xyzi, dst_label, nullptr, n_points, collapse_keep, threads);
}

int r3d_host_pack_frames_xyz(const float *const *xyzi, const uint32_t *const *label, const int32_t *n_points, int32_t B,
                             int64_t cap, float *dst_xyzi, uint32_t *dst_label, float *dst_xyz3, int32_t collapse_keep, int32_t threads) {
  if (!xyzi || !label || !n_points || !dst_xyzi || !dst_label || B <= 0 || cap <= 0)
    return r3d::fail(R3D_E_ARG, "host_pack_frames: null pointer or non-positive shape");
  for (int s = 0; s < B; ++s)
    if (n_points[s] < 0 || n_points[s] > cap || (n_points[s] > 0 && (!xyzi[s] || !label[s])))
      return r3d::fail(R3D_E_ARG, "host_pack_frames: a frame exceeds the capacity or has no data");
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  auto work = [&](int t) {
    for (int s = t; s < B; s += threads) {
      const int64_t n = n_points[s];
      stream_copy(dst_xyzi + (int64_t)s * cap * 4, xyzi[s], (size_t)n * 4 * sizeof(float));
      if (dst_xyz3) stream_xyz(dst_xyz3 + (int64_t)s * cap * 3, xyzi[s], n);
      stream_labels(dst_label + (int64_t)s * cap, label[s], n, collapse_keep);
    }
    _mm_sfence();
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  return R3D_OK;
}

// The merged cloud of every frame from the frame itself (still in the host's staging slab) and the delta the device
// exported (r3d_batch_export_delta): surviving frame points in their order, then the surviving inserted points
// (insertion.py:472-473, :526), and the check rows = every inserted point (SS tools/datasets.py:73-75, :86-88).
int r3d_host_merge_frames(const float *in_xyzi, const uint32_t *in_label, int64_t cap, const uint64_t *alive, int64_t chunks,
                          const float *tail_xyzi, const uint32_t *tail_label, int64_t tail_stride, const int32_t *counts,
                          int32_t B, float *out_xyzi, uint32_t *out_label, int64_t out_cap, int32_t *n_out, float *check,
                          int64_t check_stride, int32_t check_cols, int32_t threads) {
  if (!in_xyzi || !in_label || !alive || !tail_xyzi || !tail_label || !counts || !out_xyzi || !out_label || !n_out || B <= 0)
    return r3d::fail(R3D_E_ARG, "host_merge_frames: null pointer or no frame");
  if (check && check_cols != 4 && check_cols != 5) return r3d::fail(R3D_E_ARG, "host_merge_frames: check_cols");
  for (int s = 0; s < B; ++s) {
    const int64_t n_head = counts[s], n_total = counts[B + s];
    if (n_head < 0 || n_total < n_head || n_total > cap || n_total - n_head > tail_stride || (n_total + 63) / 64 > chunks ||
        (check && n_total - n_head > check_stride))
      return r3d::fail(R3D_E_ARG, "host_merge_frames: counts exceed the buffers");
  }
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  std::vector<int> bad(threads, 0);
  auto work = [&](int t) {
    for (int s = t; s < B; s += threads) {
      const int64_t n_head = counts[s], n_total = counts[B + s], n_tail = n_total - n_head;
      const uint64_t *aw = alive + (int64_t)s * chunks;
      const float *hx = in_xyzi + (int64_t)s * cap * 4, *tx = tail_xyzi + (int64_t)s * tail_stride * 4;
      const uint32_t *hl = in_label + (int64_t)s * cap, *tl = tail_label + (int64_t)s * tail_stride;
      float *ox = out_xyzi + (int64_t)s * out_cap * 4;
      uint32_t *ol = out_label + (int64_t)s * out_cap;
      int64_t o = 0;
      for (int64_t c = 0; c * 64 < n_total; ++c) {
        uint64_t m = aw[c];
        const int64_t base = c * 64;
        if (m == ~0ull && base + 64 <= n_head) {                                   // the usual chunk: everybody lives
          int64_t c1 = c + 1;                                                      // ... and so do its successors: one run
          while ((c1 + 1) * 64 <= n_head && aw[c1] == ~0ull) ++c1;
          const int64_t run = (c1 - c) * 64;
          if (o + run > out_cap) { bad[t] = 1; break; }
          stream_copy(ox + o * 4, hx + base * 4, (size_t)run * 4 * sizeof(float));
          stream_copy(ol + o, hl + base, (size_t)run * sizeof(uint32_t));
          o += run;
          c = c1 - 1;
          continue;
        }
        while (m) {
          const int bit = __builtin_ctzll(m);
          m &= m - 1;
          const int64_t i = base + bit;
          if (o >= out_cap) { bad[t] = 1; break; }
          if (i < n_head) {
            std::memcpy(ox + o * 4, hx + i * 4, 4 * sizeof(float));
            ol[o] = hl[i];
          } else {
            std::memcpy(ox + o * 4, tx + (i - n_head) * 4, 4 * sizeof(float));
            ol[o] = tl[i - n_head];
          }
          ++o;
        }
      }
      _mm_sfence();
      n_out[s] = (int32_t)o;
      if (check) {
        float *ck = check + (int64_t)s * check_stride * check_cols;
        for (int64_t j = 0; j < n_tail; ++j) {
          std::memcpy(ck + j * check_cols, tx + j * 4, 4 * sizeof(float));
          if (check_cols == 5) ck[j * 5 + 4] = (float)tl[j];
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  for (int t = 0; t < threads; ++t)
    if (bad[t]) return r3d::fail(R3D_E_ARG, "host_merge_frames: a merged cloud exceeds out_cap");
  return R3D_OK;
}

// ---- frames straight between files and the staging slabs (the Python file objects serialise on the interpreter lock:
// sixteen reader threads deliver fewer frames per second than one) --------------------------------------------------------
namespace {

bool read_all(int fd, void *dst, size_t bytes) {
  char *p = static_cast<char *>(dst);
  while (bytes) {
    ssize_t got = ::read(fd, p, bytes);
    if (got < 0 && errno == EINTR) continue;
    if (got <= 0) return false;
    p += got;
    bytes -= (size_t)got;
  }
  return true;
}

bool write_all(int fd, const void *src, size_t bytes) {
  const char *p = static_cast<const char *>(src);
  while (bytes) {
    ssize_t put = ::write(fd, p, bytes);
    if (put < 0 && errno == EINTR) continue;
    if (put == 0) errno = EIO;                                                    // (nothing went out and nothing is said: no progress)
    if (put <= 0) return false;
    p += put;
    bytes -= (size_t)put;
  }
  return true;
}

// Which file an I/O call failed on and why: errno is taken AT the failing call (a later close / rename / another thread's
// call would overwrite it), the temporary file is removed, and the message names the file that failed -- not the frame's
// first one.
struct IoErr {
  std::string path;
  int err = 0;
  void note(const std::string &p, int e) {
    if (path.empty()) path = p, err = e ? e : EIO;
  }
  std::string text() const { return path + " (" + std::strerror(err) + ")"; }
};

// the file whole or absent: path.tmp, then rename (what Real3DAug/tools/datasets.py:_commit does)
bool commit_file(const char *path, const void *data, size_t bytes, IoErr &e) {
  const std::string tmp = std::string(path) + ".tmp";
  int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    e.note(tmp, errno);
    return false;
  }
  bool ok = write_all(fd, data, bytes);
  if (!ok) e.note(tmp, errno);
  if (::close(fd) != 0) {
    if (ok) e.note(tmp, errno);
    ok = false;
  }
  if (ok && ::rename(tmp.c_str(), path) != 0) {
    e.note(path, errno);
    ok = false;
  }
  if (!ok) ::unlink(tmp.c_str());
  return ok;
}

}  // namespace

// velodyne/{f}.bin (float32 rows of 4) and labels/{f}.label (uint32) of B frames, the way the reference's __getitem__
// reads them (SS tools/datasets.py:51-56), straight into the staging slabs r3d_host_pack_frames fills: dst_xyzi [B][cap][4],
// dst_label [B][cap] (masked with 0xFFFF or collapsed, see there), n_points [B].  label_paths may be NULL (all labels 0).
int r3d_host_read_frames_xyz(const char *const *velodyne_paths, const char *const *label_paths, int32_t B, int64_t cap, float *dst_xyzi,
                             uint32_t *dst_label, float *dst_xyz3, int32_t *n_points, int32_t collapse_keep, int32_t threads) {
  if (!velodyne_paths || !dst_xyzi || !dst_label || !n_points || B <= 0 || cap <= 0)
    return r3d::fail(R3D_E_ARG, "host_read_frames: null pointer or non-positive shape");
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  std::vector<std::string> err(threads);
  auto work = [&](int t) {
    for (int s = t; s < B; s += threads) {
      int fd = ::open(velodyne_paths[s], O_RDONLY);
      struct stat st;
      if (fd < 0 || ::fstat(fd, &st) != 0) {
        err[t] = std::string("host_read_frames: cannot open ") + velodyne_paths[s];
        if (fd >= 0) ::close(fd);
        return;
      }
      const int64_t n = (int64_t)st.st_size / 16;
      if ((int64_t)st.st_size % 16 != 0 || n > cap) {
        err[t] = std::string("host_read_frames: not rows of 4 float32, or more points than the capacity: ") + velodyne_paths[s];
        ::close(fd);
        return;
      }
      bool ok = read_all(fd, dst_xyzi + (int64_t)s * cap * 4, (size_t)n * 16);
      ::close(fd);
      if (ok && dst_xyz3) stream_xyz(dst_xyz3 + (int64_t)s * cap * 3, dst_xyzi + (int64_t)s * cap * 4, n);
      uint32_t *dl = dst_label + (int64_t)s * cap;
      if (ok && label_paths && label_paths[s]) {
        int fl = ::open(label_paths[s], O_RDONLY);
        ok = fl >= 0 && ::fstat(fl, &st) == 0 && (int64_t)st.st_size == n * 4 && read_all(fl, dl, (size_t)n * 4);
        if (fl >= 0) ::close(fl);
        if (!ok) {
          err[t] = std::string("host_read_frames: label file missing or not one uint32 per point: ") + label_paths[s];
          return;
        }
        stream_labels(dl, dl, n, collapse_keep);
      } else if (ok) {
        std::memset(dl, 0, (size_t)n * 4);
      }
      if (!ok) {
        err[t] = std::string("host_read_frames: short read: ") + velodyne_paths[s];
        return;
      }
      n_points[s] = (int32_t)n;
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  for (int t = 0; t < threads; ++t)
    if (!err[t].empty()) return r3d::fail(R3D_E_ARG, err[t].c_str());
  return R3D_OK;
}

// The files save_data stores for B frames (SS tools/datasets.py:80-89; OD :86-93): velodyne_paths[s] <- xyzi [s][0..n_out[s])
// as float32 rows of 4, label_paths[s] (NULL entries / NULL array: not written) <- label, check_paths[s] <- check
// [s][0..n_check[s])[check_cols].  Every file under a temporary name first, then renamed; check last.
int r3d_host_write_frames(const char *const *velodyne_paths, const char *const *label_paths, const char *const *check_paths, int32_t B,
                          const float *xyzi, const uint32_t *label, int64_t cap, const int32_t *n_out, const float *check,
                          int64_t check_stride, int32_t check_cols, const int32_t *n_check, int32_t threads) {
  if (!velodyne_paths || !xyzi || !n_out || B <= 0 || cap <= 0) return r3d::fail(R3D_E_ARG, "host_write_frames: null pointer or shape");
  if (check_paths && (!check || !n_check || check_cols < 1)) return r3d::fail(R3D_E_ARG, "host_write_frames: check rows missing");
  if (label_paths && !label) return r3d::fail(R3D_E_ARG, "host_write_frames: labels missing");
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  std::vector<std::string> err(threads);
  std::vector<int> bad_arg(threads, 0);
  auto work = [&](int t) {
    for (int s = t; s < B; s += threads) {
      if (!velodyne_paths[s]) continue;                                            // (a padded slot of the last batch)
      const int64_t n = n_out[s];
      IoErr e;
      if (n < 0 || n > cap || (check_paths && check_paths[s] && (n_check[s] < 0 || n_check[s] > check_stride))) {
        err[t] = std::string("host_write_frames: counts of ") + velodyne_paths[s] + " exceed the buffers";
        bad_arg[t] = 1;
        return;
      }
      bool ok = commit_file(velodyne_paths[s], xyzi + (int64_t)s * cap * 4, (size_t)n * 16, e);
      if (ok && label_paths && label_paths[s]) ok = commit_file(label_paths[s], label + (int64_t)s * cap, (size_t)n * 4, e);
      if (ok && check_paths && check_paths[s])
        ok = commit_file(check_paths[s], check + (int64_t)s * check_stride * check_cols, (size_t)n_check[s] * check_cols * 4, e);
      if (!ok) {
        err[t] = std::string("host_write_frames: could not write ") + e.text();
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  for (int t = 0; t < threads; ++t)
    if (!err[t].empty()) return r3d::fail(bad_arg[t] ? R3D_E_ARG : R3D_E_IO, err[t].c_str());
  return R3D_OK;
}

// The files of B frames straight from what the host already holds -- the frames in the staging slab the upload was made
// from, and the delta the device exported (r3d_batch_export_delta) -- without materialising the merged clouds first:
// velodyne/{f}.bin = the surviving frame points, run by run (writev over the staging slab), then the surviving inserted
// points; labels/{f}.label the same over the label slab; check/{f}.bin = every inserted point (SS tools/datasets.py:73-75,
// :80-89; OD :86-93).  The same bytes r3d_host_merge_frames + r3d_host_write_frames write, at half the host's memory traffic
// (the merge read and wrote every frame once more before the writers read it again: the file legs are bound by exactly that).
}  // extern "C"

namespace {

// runs of set bits of a frame's alive words over [lo, hi), as (first point, count); `emit` returns false to stop
template <class Emit>
bool alive_runs(const uint64_t *aw, int64_t lo, int64_t hi, Emit &&emit) {
  int64_t run0 = -1;
  for (int64_t c = lo >> 6; (c << 6) < hi; ++c) {
    uint64_t m = aw[c];
    const int64_t base = c << 6;
    if (base < lo) m &= ~0ull << (lo - base);
    if (base + 64 > hi) m &= (hi - base) >= 64 ? ~0ull : ((1ull << (hi - base)) - 1ull);
    if (m == ~0ull) {                                                  // the usual word: the run goes on
      if (run0 < 0) run0 = base;
      continue;
    }
    int64_t at = base;                                                 // next point of the word not looked at yet
    while (true) {
      if (run0 >= 0) {                                                 // inside a run: where does it end?
        const uint64_t rest = ~m & (at - base >= 64 ? 0ull : (~0ull << (at - base)));
        if (!rest) break;                                              // ... not in this word
        const int64_t end = base + __builtin_ctzll(rest);
        if (!emit(run0, end - run0)) return false;
        run0 = -1;
        at = end;
      } else {                                                         // outside: where does the next one start?
        const uint64_t rest = m & (at - base >= 64 ? 0ull : (~0ull << (at - base)));
        if (!rest) break;
        run0 = base + __builtin_ctzll(rest);
        at = run0;
      }
    }
  }
  if (run0 >= 0) {
    const int64_t end = hi;
    if (end > run0 && !emit(run0, end - run0)) return false;
  }
  return true;
}

struct IovFile {
  int fd = -1;
  std::vector<iovec> iov;
  bool ok = true;
  int err = 0;                       // errno of the writev that failed
  std::string tmp;
  bool flush() {
    size_t done = 0;
    while (ok && done < iov.size()) {
      const int n = (int)std::min<size_t>(iov.size() - done, IOV_MAX);
      ssize_t want = 0;
      for (int i = 0; i < n; ++i) want += (ssize_t)iov[done + i].iov_len;
      ssize_t put = ::writev(fd, iov.data() + done, n);
      if (put < 0 && errno == EINTR) continue;
      if (put == want) {
        done += (size_t)n;
        continue;
      }
      if (put <= 0) {                                                    // (0 with bytes outstanding: no progress, not a retry)
        err = put < 0 ? errno : EIO;
        ok = false;
        break;
      }
      // a short write: skip what went out, go on from there
      size_t i = done;
      while (put > 0 && (size_t)put >= iov[i].iov_len) put -= (ssize_t)iov[i++].iov_len;
      if (put > 0) {
        iov[i].iov_base = static_cast<char *>(iov[i].iov_base) + put;
        iov[i].iov_len -= (size_t)put;
      }
      done = i;
    }
    iov.clear();
    return ok;
  }
  void add(const void *p, size_t bytes) {
    if (!bytes) return;
    if (!iov.empty() && static_cast<char *>(iov.back().iov_base) + iov.back().iov_len == p) iov.back().iov_len += bytes;
    else iov.push_back(iovec{const_cast<void *>(p), bytes});
    if (iov.size() >= 4096) flush();
  }
};

bool open_tmp(const char *path, IovFile &f, IoErr &e) {
  f.tmp = std::string(path) + ".tmp";
  f.fd = ::open(f.tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (f.fd < 0) e.note(f.tmp, errno);
  return f.fd >= 0;
}
bool close_commit(const char *path, IovFile &f, IoErr &e) {
  bool ok = f.flush();
  if (!ok) e.note(f.tmp, f.err);
  if (::close(f.fd) != 0) {
    if (ok) e.note(f.tmp, errno);
    ok = false;
  }
  f.fd = -1;
  if (ok && ::rename(f.tmp.c_str(), path) != 0) {
    e.note(path, errno);
    ok = false;
  }
  if (!ok) ::unlink(f.tmp.c_str());
  return ok;
}

}  // namespace

extern "C" {

int r3d_host_write_delta_frames(const char *const *velodyne_paths, const char *const *label_paths, const char *const *check_paths,
                                int32_t B, const float *in_xyzi, const uint32_t *in_label, int64_t cap, const uint64_t *alive,
                                int64_t chunks, const float *tail_xyzi, const uint32_t *tail_label, int64_t tail_stride,
                                const int32_t *counts, int32_t check_cols, int32_t *n_out, int32_t threads) {
  if (!velodyne_paths || !in_xyzi || !alive || !tail_xyzi || !tail_label || !counts || B <= 0 || cap <= 0)
    return r3d::fail(R3D_E_ARG, "host_write_delta_frames: null pointer or shape");
  if (label_paths && !in_label) return r3d::fail(R3D_E_ARG, "host_write_delta_frames: labels missing");
  if (check_paths && check_cols != 4 && check_cols != 5) return r3d::fail(R3D_E_ARG, "host_write_delta_frames: check_cols");
  for (int s = 0; s < B; ++s) {
    const int64_t n_head = counts[s], n_total = counts[B + s];
    if (n_head < 0 || n_total < n_head || n_total > cap || n_total - n_head > tail_stride || (n_total + 63) / 64 > chunks)
      return r3d::fail(R3D_E_ARG, "host_write_delta_frames: counts exceed the buffers");
  }
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  std::vector<std::string> err(threads);
  auto work = [&](int t) {
    std::vector<float> ck5;
    for (int s = t; s < B; s += threads) {
      const int64_t n_head = counts[s], n_total = counts[B + s], n_tail = n_total - n_head;
      const uint64_t *aw = alive + (int64_t)s * chunks;
      const float *hx = in_xyzi + (int64_t)s * cap * 4, *tx = tail_xyzi + (int64_t)s * tail_stride * 4;
      const uint32_t *hl = in_label ? in_label + (int64_t)s * cap : nullptr, *tl = tail_label + (int64_t)s * tail_stride;
      int64_t survivors = 0;
      alive_runs(aw, 0, n_total, [&](int64_t, int64_t n) { survivors += n; return true; });
      if (n_out) n_out[s] = (int32_t)survivors;
      if (!velodyne_paths[s]) continue;                                            // (a padded slot of the last batch)
      bool ok = true;
      IoErr e;
      {
        IovFile f;
        ok = open_tmp(velodyne_paths[s], f, e);
        if (ok) {
          alive_runs(aw, 0, n_head, [&](int64_t i, int64_t n) { f.add(hx + i * 4, (size_t)n * 16); return f.ok; });
          alive_runs(aw, n_head, n_total, [&](int64_t i, int64_t n) { f.add(tx + (i - n_head) * 4, (size_t)n * 16); return f.ok; });
          ok = close_commit(velodyne_paths[s], f, e);
        }
      }
      if (ok && label_paths && label_paths[s]) {
        IovFile f;
        ok = open_tmp(label_paths[s], f, e);
        if (ok) {
          alive_runs(aw, 0, n_head, [&](int64_t i, int64_t n) { f.add(hl + i, (size_t)n * 4); return f.ok; });
          alive_runs(aw, n_head, n_total, [&](int64_t i, int64_t n) { f.add(tl + (i - n_head), (size_t)n * 4); return f.ok; });
          ok = close_commit(label_paths[s], f, e);
        }
      }
      if (ok && check_paths && check_paths[s]) {
        const void *data = tx;
        if (check_cols == 5) {
          ck5.resize((size_t)n_tail * 5);
          for (int64_t j = 0; j < n_tail; ++j) {
            std::memcpy(ck5.data() + j * 5, tx + j * 4, 4 * sizeof(float));
            ck5[(size_t)j * 5 + 4] = (float)tl[j];
          }
          data = ck5.data();
        }
        ok = commit_file(check_paths[s], data, (size_t)n_tail * check_cols * 4, e);
      }
      if (!ok) {
        err[t] = std::string("host_write_delta_frames: could not write ") + e.text();
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  for (int t = 0; t < threads; ++t)
    if (!err[t].empty()) return r3d::fail(R3D_E_IO, err[t].c_str());
  return R3D_OK;
}


// OD tools/datasets.py:20-37 (create_annotation) for the frames of a batch: dst[i] = the bytes of src[i] followed by extra[i]
// (the lines of the inserted objects), written to dst[i].tmp and renamed.  A null dst[i] is skipped, a null extra[i] adds
// nothing.  One frame per thread at a time (a Python loop of 4 096 frames took 0.3 s of a 0.6 s file-to-file run).
int r3d_host_append_text_files(const char *const *src, const char *const *dst, const char *const *extra, int32_t n,
                               int32_t threads) {
  if (!src || !dst || !extra || n < 0) return r3d::fail(R3D_E_ARG, "host_append_text_files: null pointer or count");
  if (n == 0) return R3D_OK;
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  std::vector<std::string> err(threads);
  std::vector<int> bad_arg(threads, 0);
  auto work = [&](int t) {
    std::string text;
    for (int i = t; i < n; i += threads) {
      if (!dst[i]) continue;
      if (!src[i]) {
        err[t] = "host_append_text_files: a frame without its source file";
        bad_arg[t] = 1;
        return;
      }
      text.clear();
      bool ok = true;
      IoErr e;
      {
        const int fd = ::open(src[i], O_RDONLY);
        ok = fd >= 0;
        if (!ok) e.note(src[i], errno);
        char buf[4096];
        while (ok) {
          const ssize_t got = ::read(fd, buf, sizeof buf);
          if (got < 0 && errno == EINTR) continue;
          if (got < 0) {
            e.note(src[i], errno);
            ok = false;
          }
          if (got <= 0) break;
          text.append(buf, (size_t)got);
        }
        if (fd >= 0) ::close(fd);
      }
      if (ok) {
        // the reference reads the source in text mode (OD tools/datasets.py:23-27): "\r\n" and a lone "\r" arrive as "\n"
        // (universal newlines), and are written back as "\n"
        size_t o = 0;
        for (size_t k = 0; k < text.size(); ++k) {
          char c = text[k];
          if (c == '\r') {
            if (k + 1 < text.size() && text[k + 1] == '\n') ++k;
            c = '\n';
          }
          text[o++] = c;
        }
        text.resize(o);
        if (extra[i]) text += extra[i];
        ok = commit_file(dst[i], text.data(), text.size(), e);
      }
      if (!ok) {
        err[t] = std::string("host_append_text_files: ") + src[i] + " -> " + dst[i] + ": " + e.text();
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  for (int t = 0; t < threads; ++t)
    if (!err[t].empty()) return r3d::fail(bad_arg[t] ? R3D_E_ARG : R3D_E_IO, err[t].c_str());
  return R3D_OK;
}

}  // extern "C"
