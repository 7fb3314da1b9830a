// Host side of the file-to-file pipeline: the packer that lays the frames of
// a batch out in the pinned staging buffer the way r3d_batch_t wants them on the device.  Plain C++
// (std::thread), no HIP: the copies into HBM are the caller's (hipMemcpyAsync on its copy streams).
#include <cstring>
#include <thread>
#include <vector>

#include "r3d_host.hpp"

extern "C" {

int r3d_host_pack_frames(const float *const *xyzi, const uint32_t *const *label, const int32_t *n_points, int32_t B,
                         int64_t cap, float *dst_xyzi, uint32_t *dst_label, int32_t collapse_keep, int32_t threads) {
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
      std::memcpy(dst_xyzi + (int64_t)s * cap * 4, xyzi[s], (size_t)n * 4 * sizeof(float));
      uint32_t *dl = dst_label + (int64_t)s * cap;
      const uint32_t *sl = label[s];
      if (collapse_keep < 0)
        for (int64_t i = 0; i < n; ++i) dl[i] = sl[i] & 0xFFFFu;                   // SS tools/datasets.py:53-55
      else                                                                         // OD insertion.py:353-355
        for (int64_t i = 0; i < n; ++i) dl[i] = (sl[i] & 0xFFFFu) == (uint32_t)collapse_keep ? (uint32_t)collapse_keep : 1u;
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  return R3D_OK;
}

}  // extern "C"
