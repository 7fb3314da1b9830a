"""Frames streamed through the batched hot path with the transfers overlapped.

``StreamedAugmenter`` keeps a few *lanes*.  A lane is one ``SceneBatch`` on the device plus pinned
host buffers for everything that crosses the PCIe link and its own HIP stream.  ``submit`` packs a batch
of frames into the lane's pinned input (the native packer ``r3d_host_pack_frames``: threads, no Python loop
over points), then enqueues, on the lane's stream, upload -> ``begin`` -> ``insert_many`` -> export of the
DELTA -> download, and returns at once; ``collect`` waits for the lane's event, merges on the host and hands
out views of the lane's output buffers.  With two or more lanes in flight the upload of batch i+1 and the
download of batch i-1 run on the copy engines while the kernels of batch i run on the CUs, and the host packs
and merges meanwhile.

What comes back from the device is the delta, not the merged cloud (99.4 % of which is the frame the host
still holds in the lane's pinned input): one alive bit per point, the inserted points, four counters per
frame -- 0.1 MB per frame instead of 2.5 MB.  ``r3d_host_merge_frames`` (C++ threads) writes the merged cloud,
the labels and the check rows from that: the bytes of ``velodyne/ labels/ check/`` (SS tools/datasets.py:72-91).
``delta=False`` keeps the round-2 behaviour (``r3d_batch_finish`` on the device, whole clouds downloaded).

One placement candidate per insert slot (what ``r3d_batch_insert_many`` takes); the candidate loop
with several placements per slot stays with ``AugmentPipeline.run`` / ``run_placed``.
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _lib
from .batch import SceneBatch


import os as _os
_SUM_COUNTERS = {} if _os.environ.get("R3D_SUM_COUNTERS") else None      # diagnostics (tools/soak_files.py)


class _Lane:
    def __init__(self, B, cap, log_cap, K, sample_rows, rows, cols, device, check_cols, delta, xyz_upload=False):
        torch = _lib.require_gpu()
        self.torch = torch
        self.bt = SceneBatch(B, cap, log_cap, rows=rows, cols=cols, device=device)
        self.B, self.K, self.check_cols = B, K, check_cols
        cap, log_cap = self.bt.cap, self.bt.log_cap
        self.chunks = (cap + 63) // 64
        with _lib.on(device):
            self.stream = torch.cuda.Stream()
            self.copy_streams = [torch.cuda.Stream() for _ in range(4)]
            self.done = torch.cuda.Event()
            pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            self.in_xyzi, self.in_label, self.in_n = pin((B, cap, 4), torch.float32), pin((B, cap), torch.int32), pin((B,), torch.int32)
            self.in_rows = [pin((sample_rows, 5), torch.float64) for _ in range(K)]
            self.in_off = pin((max(K, 1), B + 1), torch.int64)
            self.in_need = pin((max(K, 1), B), torch.int32)
            self.d_rows = [torch.empty((sample_rows, 5), dtype=torch.float64, device=device) for _ in range(K)]
            self.d_off = torch.zeros((max(K, 1), B + 1), dtype=torch.int64, device=device)
            self.d_need = torch.zeros((max(K, 1), B), dtype=torch.int32, device=device)
            self.out_counts = pin((5, B), torch.int32)                  # n_out, n_log, status, rebases, points numbered anew
            self.out_acc = pin((max(K, 1), B), torch.int32)
            if delta:
                # xyz_upload: what crosses the link of a frame is x y z alone, 12 bytes per point (r3d_batch_begin_xyz) -- the
                # host packer writes them beside the full rows, which stay on the host for the merge / the file writers
                self.in_xyz3 = pin((B, cap, 3), torch.float32) if xyz_upload else None
                self.d_xyz3 = torch.empty((B, cap, 3), dtype=torch.float32, device=device) if xyz_upload else None
                # the delta on the device and in pinned memory; the merged results in plain host memory
                self.d_alive = torch.zeros((B, self.chunks), dtype=torch.int64, device=device)
                self.d_tail_xyzi = torch.zeros((B, log_cap, 4), dtype=torch.float32, device=device)
                self.d_tail_label = torch.zeros((B, log_cap), dtype=torch.int32, device=device)
                self.d_dcounts = torch.zeros((2, B), dtype=torch.int32, device=device)
                self.h_alive, self.h_tail_xyzi = pin((B, self.chunks), torch.int64), pin((B, log_cap, 4), torch.float32)
                self.h_tail_label, self.h_dcounts = pin((B, log_cap), torch.int32), pin((2, B), torch.int32)
                self.out_xyzi = torch.empty((B, cap, 4), dtype=torch.float32)
                self.out_label = torch.empty((B, cap), dtype=torch.int32)
                self.out_check = torch.empty((B, log_cap, max(check_cols, 4)), dtype=torch.float32)
                self.h_n_out = torch.zeros((B,), dtype=torch.int32)
            else:
                self.out_xyzi, self.out_label = pin((B, cap, 4), torch.float32), pin((B, cap), torch.int32)
                self.out_check = pin((B, log_cap, max(check_cols, 4)), torch.float32)
            # the batch was built on the device's current stream: this lane's stream starts behind it
            self.stream.wait_stream(torch.cuda.current_stream())
            # the row counts of the check files as `collect` saw them: the pinned out_counts are overwritten by the
            # lane's next download, the snapshot belongs to whoever consumes the lane's results
            self.h_n_log = torch.zeros((B,), dtype=torch.int32)
        self.busy = False
        self.tag = None
        if os.environ.get("R3D_DUMP_BUFFERS"):                   # diagnostic: where every buffer of the lane lives
            import sys
            for owner, obj in (("lane", self), ("batch", self.bt)):
                for name, v in sorted(vars(obj).items()):
                    for j, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
                        if hasattr(t, "data_ptr") and hasattr(t, "numel") and t.numel():
                            a, nb = t.data_ptr(), t.numel() * t.element_size()
                            print(f"BUF {owner}.{name}[{j}] {a:#x} .. {a + nb:#x} ({nb} B, {'device' if t.is_cuda else 'host'})", file=sys.stderr, flush=True)


class StreamedAugmenter:
    def __init__(self, B, n_max, grow, n_slots, sample_rows, lanes=3, rows=_lib.NUMROW, cols=_lib.NUMCOLUMN,
                 device="cuda:0", check_cols=5, collapse_keep=-1, pack_threads=16, delta=True, xyz_upload=False):
        """B frames per batch, at most n_max points per frame, `grow` inserted points per frame in all
        (sum over the slots), n_slots insert slots, at most sample_rows sample points per slot and batch."""
        self.lib = _lib.load()
        self.B, self.K = int(B), int(n_slots)
        self.collapse_keep, self.pack_threads, self.delta = int(collapse_keep), int(pack_threads), bool(delta)
        # delta mode: upload x y z alone (12 bytes per point instead of 16).  Off by default: on the boxes measured the streamed
        # legs are bound by the HOST's memory traffic, not by the link, and the second staging slab adds 1.4 MB of stores per
        # frame for 0.5 MB less over the link (round 6: 17.8 -> 14.2 thousand frames/s)
        self.xyz_upload = bool(xyz_upload) and self.delta
        self.check_cols = int(check_cols)
        import os
        self.copy_streams = max(1, min(4, int(os.environ.get("R3D_COPY_STREAMS", "1"))))
        self.lanes = [_Lane(B, n_max + grow, max(grow, 1), self.K, max(int(sample_rows), 1), rows, cols, device, self.check_cols,
                            self.delta, self.xyz_upload) for _ in range(lanes)]
        self.device = device
        # delta mode: False = `collect` leaves the merged clouds unmade (results come back as None) for a caller that only
        # writes files -- `write_files` then writes them straight from the lane's pinned input and the delta, run by run of
        # surviving points (r3d_host_write_delta_frames: no merged copy in between); AugmentPipeline.run_streamed does that
        self.merge_on_collect = True
        self.bytes_h2d = self.bytes_d2h = 0
        # seconds spent per stage, summed over the batches (submitting thread: read / pack, enqueue; drain thread: wait for
        # the device, host merge, the caller's consume)
        self.times = {"read_or_pack": 0.0, "enqueue": 0.0, "wait_device": 0.0, "merge": 0.0, "consume": 0.0, "wait_free_lane": 0.0}

    def free_lane(self):
        for i, ln in enumerate(self.lanes):
            if not ln.busy:
                return i
        return None

    def submit(self, lane_no, scenes, inserts, min_points, tag=None):
        """scenes: B x (xyzi float32 [n,4], label uint32 [n]) host arrays as read from the files;
        inserts[s][k]: M x 5 float64 sample of slot k of frame s (or None; a frame may have fewer than n_slots
        inserts); min_points[s][k].  Returns immediately; the lane is busy until ``collect``."""
        ln = self.lanes[lane_no]
        assert not ln.busy and len(scenes) == self.B
        torch, bt, B, K = ln.torch, ln.bt, self.B, self.K
        # -- pack on the host (pinned staging); the frames by the native packer
        xs = [np.ascontiguousarray(x, dtype=np.float32) for x, _ in scenes]
        ls = [np.ascontiguousarray(l, dtype=np.uint32) for _, l in scenes]
        n = np.array([len(x) for x in xs], dtype=np.int32)
        px = (C.c_void_p * B)(*[x.ctypes.data for x in xs])
        pl = (C.c_void_p * B)(*[l.ctypes.data for l in ls])
        import time
        t0 = time.perf_counter()
        _lib.check(self.lib.r3d_host_pack_frames_xyz(px, pl, n.ctypes.data, B, bt.cap, ln.in_xyzi.data_ptr(), ln.in_label.data_ptr(),
                                                     ln.in_xyz3.data_ptr() if self.xyz_upload else None, self.collapse_keep,
                                                     self.pack_threads), "r3d_host_pack_frames_xyz")
        ln.in_n.numpy()[:] = n
        self.times["read_or_pack"] += time.perf_counter() - t0
        return self._enqueue(lane_no, inserts, min_points, tag)

    def submit_files(self, lane_no, velodyne_files, label_files, inserts, min_points, tag=None):
        """Like ``submit`` with the frames still in their files (velodyne/{f}.bin, labels/{f}.label; label_files
        may be None): native threads read them straight into the lane's pinned input (``r3d_host_read_frames``)."""
        ln = self.lanes[lane_no]
        assert not ln.busy and len(velodyne_files) == self.B
        B = self.B
        enc = lambda paths: (C.c_char_p * B)(*[None if p is None else str(p).encode() for p in paths])
        pv = enc(velodyne_files)
        pl = enc(label_files) if label_files is not None else None
        import time
        t0 = time.perf_counter()
        _lib.check(self.lib.r3d_host_read_frames_xyz(pv, pl, B, ln.bt.cap, ln.in_xyzi.data_ptr(), ln.in_label.data_ptr(),
                                                     ln.in_xyz3.data_ptr() if self.xyz_upload else None, ln.in_n.data_ptr(),
                                                     self.collapse_keep, self.pack_threads), "r3d_host_read_frames_xyz")
        self.times["read_or_pack"] += time.perf_counter() - t0
        return self._enqueue(lane_no, inserts, min_points, tag)

    def write_files(self, lane_no, velodyne_files, label_files, check_files):
        """The results of the lane (after ``collect``, before its next submit) into files, by native threads
        (``r3d_host_write_frames``): lists of B paths, None = skip (label_files / check_files may be None altogether)."""
        ln = self.lanes[lane_no]
        B = self.B
        enc = lambda paths: None if paths is None else (C.c_char_p * B)(*[None if p is None else str(p).encode() for p in paths])
        if self.delta and not getattr(ln, "merged", True):
            bt = ln.bt
            _lib.check(self.lib.r3d_host_write_delta_frames(
                enc(velodyne_files), enc(label_files), enc(check_files) if ln.check_cols else None, B, ln.in_xyzi.data_ptr(),
                ln.in_label.data_ptr(), bt.cap, ln.h_alive.data_ptr(), ln.chunks, ln.h_tail_xyzi.data_ptr(), ln.h_tail_label.data_ptr(),
                bt.log_cap, ln.h_dcounts.data_ptr(), ln.check_cols or 4, ln.h_n_out.data_ptr(), self.pack_threads),
                "r3d_host_write_delta_frames")
            return
        n_out = ln.h_n_out if self.delta else ln.out_counts[0]
        cc = max(ln.check_cols, 4)
        _lib.check(self.lib.r3d_host_write_frames(
            enc(velodyne_files), enc(label_files), enc(check_files) if ln.check_cols else None, B, ln.out_xyzi.data_ptr(),
            ln.out_label.data_ptr(), ln.bt.cap, n_out.data_ptr(), ln.out_check.data_ptr(), ln.bt.log_cap, cc,
            ln.h_n_log.data_ptr(), self.pack_threads), "r3d_host_write_frames")

    def _enqueue(self, lane_no, inserts, min_points, tag):
        import time
        t_enq = time.perf_counter()
        ln = self.lanes[lane_no]
        torch, bt, B, K = ln.torch, ln.bt, self.B, self.K
        off = ln.in_off.numpy()
        need = ln.in_need.numpy()
        empty = np.empty((0, 5))
        for k in range(K):
            # one concatenation per slot (a Python loop over B x K slices costs 4 ms per 256 frames)
            parts = [inserts[s][k] if k < len(inserts[s]) and inserts[s][k] is not None and len(inserts[s][k]) else empty
                     for s in range(B)]
            lens = np.fromiter((len(p) for p in parts), dtype=np.int64, count=B)
            off[k, 0] = 0
            np.cumsum(lens, out=off[k, 1:])
            m = int(off[k, B])
            if m > len(ln.in_rows[k]):
                raise ValueError(f"slot {k}: {m} sample points in the batch, the lane was sized for {len(ln.in_rows[k])}")
            if m:
                np.concatenate(parts, axis=0, out=ln.in_rows[k].numpy()[:m])
            need[k] = [min_points[s][k] if k < len(min_points[s]) else 0 for s in range(B)]
        # -- everything else on the lane's stream: upload, kernels, download
        with _lib.on(self.device), torch.cuda.stream(ln.stream):
            if self.copy_streams > 1:
                # the frames in pieces on several streams: one hipMemcpyAsync keeps ONE copy engine busy
                # (~36 GB/s on this box), pieces on several streams are taken by several engines
                ev = torch.cuda.Event()
                ev.record(ln.stream)
                n_piece = self.copy_streams
                step = (B + n_piece - 1) // n_piece
                for j, cs in enumerate(ln.copy_streams[:n_piece]):
                    lo, hi = j * step, min(B, (j + 1) * step)
                    if lo >= hi:
                        continue
                    cs.wait_event(ev)
                    with torch.cuda.stream(cs):
                        if self.xyz_upload:
                            ln.d_xyz3[lo:hi].copy_(ln.in_xyz3[lo:hi], non_blocking=True)
                        else:
                            bt.xyzi[lo:hi].copy_(ln.in_xyzi[lo:hi], non_blocking=True)
                        if not self.delta:
                            bt.label[lo:hi].copy_(ln.in_label[lo:hi], non_blocking=True)
                    ln.stream.wait_stream(cs)
            else:
                if self.xyz_upload:
                    ln.d_xyz3.copy_(ln.in_xyz3, non_blocking=True)   # x y z only: 12 bytes per point
                else:
                    bt.xyzi.copy_(ln.in_xyzi, non_blocking=True)     # whole slabs: one contiguous copy each
                if not self.delta:
                    # the frames' labels are read by r3d_batch_finish only: in delta mode they stay on the host,
                    # where the merge takes them from the staging slab (the device holds the inserted points' labels)
                    bt.label.copy_(ln.in_label, non_blocking=True)
            bt.n_points.copy_(ln.in_n, non_blocking=True)
            for k in range(K):
                m = int(off[k, B])
                if m:
                    ln.d_rows[k][:m].copy_(ln.in_rows[k][:m], non_blocking=True)
                self.bytes_h2d += m * 40
            ln.d_off.copy_(ln.in_off, non_blocking=True)
            ln.d_need.copy_(ln.in_need, non_blocking=True)
            self.bytes_h2d += B * bt.cap * (12 if self.xyz_upload else (16 if self.delta else 20))
            if self.xyz_upload:
                bt.begin_xyz(ln.d_xyz3)
            else:
                bt.begin()
            if K:
                _, acc = bt.insert_many_device([(ln.d_rows[k], ln.d_off[k]) for k in range(K)], [ln.d_need[k] for k in range(K)])
                ln.out_acc.copy_(acc, non_blocking=True)
            if self.delta:
                _lib.check(self.lib.r3d_batch_export_delta(C.byref(bt.desc), ln.d_alive.data_ptr(), ln.d_tail_xyzi.data_ptr(),
                                                           ln.d_tail_label.data_ptr(), bt.log_cap, ln.d_dcounts.data_ptr(),
                                                           _lib.stream_ptr()), "r3d_batch_export_delta")
                ln.h_alive.copy_(ln.d_alive, non_blocking=True)
                ln.h_tail_xyzi.copy_(ln.d_tail_xyzi, non_blocking=True)
                ln.h_tail_label.copy_(ln.d_tail_label, non_blocking=True)
                ln.h_dcounts.copy_(ln.d_dcounts, non_blocking=True)
                self.bytes_d2h += ln.h_alive.numel() * 8 + ln.h_tail_xyzi.numel() * 4 + ln.h_tail_label.numel() * 4
            else:
                bt.finish(ln.check_cols)
                ln.out_xyzi.copy_(bt.out_xyzi, non_blocking=True)
                ln.out_label.copy_(bt.out_label, non_blocking=True)
                if bt.check is not None:
                    ln.out_check[:, :, :ln.check_cols].copy_(bt.check, non_blocking=True)
                ln.out_counts[0].copy_(bt.n_out, non_blocking=True)
                self.bytes_d2h += B * bt.cap * 20 + ln.out_check.numel() * 4
            ln.out_counts[1].copy_(bt.n_log, non_blocking=True)
            ln.out_counts[2].copy_(bt.status, non_blocking=True)
            ln.out_counts[3].copy_(bt.rebase, non_blocking=True)
            ln.looked = bt._looked                                  # (did this begin look at the point order?  SceneBatch.order)
            if ln.looked:
                ln.out_counts[4].copy_(bt.point_order_device(), non_blocking=True)
            ln.done.record(ln.stream)
        ln.busy, ln.tag = True, tag
        self.times["enqueue"] += time.perf_counter() - t_enq
        return lane_no

    def collect(self, lane_no):
        """Wait for the lane; (tag, results, accepted): results[s] = (xyzi [n,4] float32, label [n] uint32,
        check [m,cols] float32) as VIEWS of the lane's buffers (valid until the lane's next submit),
        accepted[s][k] = 0 (accepted) or -1."""
        self.wait(lane_no)
        return self.finish_collect(lane_no)

    def wait(self, lane_no):
        """The first half of ``collect``: block until the lane's device work and downloads are done (``run`` does this on a
        thread of its own, so that the host merge of one lane overlaps the wait for the next)."""
        import time
        ln = self.lanes[lane_no]
        assert ln.busy
        t0 = time.perf_counter()
        ln.done.synchronize()
        self.times["wait_device"] += time.perf_counter() - t0

    def finish_collect(self, lane_no):
        """The second half of ``collect`` (after ``wait``): status check, host merge, the results' views."""
        import time
        ln = self.lanes[lane_no]
        assert ln.busy
        t1 = time.perf_counter()
        counts = ln.out_counts.numpy()
        if _SUM_COUNTERS is not None:                                # R3D_SUM_COUNTERS=1: the insert kernels' diagnostic counters, summed
            for k, v in ln.bt.debug_counters(reset=True).items():
                _SUM_COUNTERS[k] = (_SUM_COUNTERS.get(k, 0) + v) if not isinstance(v, list) else [a + b for a, b in zip(_SUM_COUNTERS.get(k, [0] * len(v)), v)]
            _SUM_COUNTERS["rebases"] = _SUM_COUNTERS.get("rebases", 0) + int(counts[3].sum())
        if getattr(ln, "looked", False):
            ln.bt.note_point_order(counts[4])
        n_log = ln.h_n_log.numpy()
        n_log[:] = counts[1]
        redo = [int(s) for s in np.nonzero(counts[2])[0] if _lib.needs_level1(counts[2][s])]   # see _redo_level1
        for s in np.nonzero(counts[2])[0]:
            if int(s) in redo:
                continue
            ln.busy = False
            try:
                why = ln.bt.debug_counters(reset=False)
            except Exception as e:                                  # (the device is gone: keep the status)
                why = repr(e)[:80]
            _lib.raise_status(int(counts[2][s]), f"scene {s} of the batch (status {int(counts[2][s])}, {int(counts[3][s])} rebases, "
                                                 f"{int(counts[1][s])} inserted points; counters {why})")
        cc = max(ln.check_cols, 4)
        ln.merged = True
        if self.delta and not self.merge_on_collect and not redo:
            # (the files are written from the pinned input and the delta: write_files)
            ln.merged = False
            acc = ln.out_acc.numpy()
            accepted = [[0 if acc[k, s] else -1 for k in range(self.K)] for s in range(self.B)]
            ln.busy = False
            return ln.tag, [None] * self.B, accepted
        if self.delta:
            bt = ln.bt
            _lib.check(self.lib.r3d_host_merge_frames(
                ln.in_xyzi.data_ptr(), ln.in_label.data_ptr(), bt.cap, ln.h_alive.data_ptr(), ln.chunks, ln.h_tail_xyzi.data_ptr(),
                ln.h_tail_label.data_ptr(), bt.log_cap, ln.h_dcounts.data_ptr(), self.B, ln.out_xyzi.data_ptr(),
                ln.out_label.data_ptr(), bt.cap, ln.h_n_out.data_ptr(), ln.out_check.data_ptr() if ln.check_cols else None,
                bt.log_cap, cc, self.pack_threads), "r3d_host_merge_frames")
            n_out = ln.h_n_out.numpy()
            self.times["merge"] += time.perf_counter() - t1
        else:
            n_out = counts[0]
        if redo:
            try:
                for s in redo:
                    self._redo_level1(ln, s, n_out, n_log)
            except Exception:
                ln.busy = False
                raise
        ox, ol, ck = ln.out_xyzi.numpy(), ln.out_label.numpy().view(np.uint32), ln.out_check.numpy()
        acc = ln.out_acc.numpy()
        results = [(ox[s, :n_out[s]], ol[s, :n_out[s]], ck[s, :n_log[s], :ln.check_cols] if ln.check_cols else None)
                   for s in range(self.B)]
        accepted = [[0 if acc[k, s] else -1 for k in range(self.K)] for s in range(self.B)]
        ln.busy = False
        return ln.tag, results, accepted

    def _redo_level1(self, ln, s, n_out, n_log):
        """Frame s of the lane came back beyond the batched kernels' limits (``_lib.S_REDO_LEVEL1``: an insert's window exceeds
        a CU's LDS -- an object a few metres from the sensor on a grid several times the reference's --, a sample of more
        than R3D_MAX_SAMPLE points, more than R3D_FAR_CAP pixels beyond 500 m): once more, alone, through the Level-1
        kernels (``level1.augment_scene``), its results into the lane's output slabs."""
        from . import level1
        bt, K = ln.bt, self.K
        n = int(ln.in_n.numpy()[s])
        off, need = ln.in_off.numpy(), ln.in_need.numpy()
        cands = [[ln.in_rows[k].numpy()[int(off[k, s]):int(off[k, s + 1])].copy()] for k in range(K)]
        (x, l, ck), acc = level1.augment_scene(ln.in_xyzi.numpy()[s, :n], ln.in_label.numpy().view(np.uint32)[s, :n], cands,
                                               [int(need[k, s]) for k in range(K)], bt.rows, bt.cols, self.device, ln.check_cols)
        ln.out_xyzi.numpy()[s, :len(x)] = x
        ln.out_label.numpy().view(np.uint32)[s, :len(l)] = l
        n_out[s] = len(x)
        if ln.check_cols:
            ln.out_check.numpy()[s, :len(ck), :ln.check_cols] = ck
            n_log[s] = len(ck)
        for k in range(K):
            ln.out_acc.numpy()[k, s] = 1 if acc[k] >= 0 else 0
        self.level1_frames = getattr(self, "level1_frames", 0) + 1

    def run(self, batches, consume):
        """batches: iterable of (scenes, inserts, min_points, tag); consume(tag, results, accepted) is
        called in submission order while later batches are in flight.  This thread packs and submits; a second
        one waits for the lanes in order, merges (delta mode) and calls ``consume`` -- so the host's packing of
        batch i+1 and its merging / consuming of batch i-1 overlap as well."""
        import queue
        import threading
        import time
        submitted, merged, free, errors = queue.Queue(), queue.Queue(), queue.Queue(), []
        # a lane goes back to the submitting thread only when `consume` has returned: its buffers (results, counters,
        # the pinned input the host merge reads) belong to the drain thread until then.  `collect` clears `busy`
        # earlier, so lanes are taken from this queue and never from free_lane().
        for i in range(len(self.lanes)):
            free.put(i)

        # four stages on four threads: this one packs / reads and submits; `waiter` waits for the lanes' device work in
        # order; `drain` checks the status and merges (delta mode); `hand_over` calls consume -- file writing, as a rule --
        # and gives the lane back.  (Merge and consume on ONE thread were the bottleneck of the file-to-file legs: 0.28 +
        # 0.33 s per 4 096 frames; round 6: the wait for the device and the merge on one thread were the in-memory leg's,
        # 0.07 + 0.14 s per 4 096 frames beside 0.17 s of packing and enqueueing.)
        waited = queue.Queue()

        def waiter():
            while True:
                lane = submitted.get()
                if lane is None:
                    waited.put(None)
                    return
                try:
                    if not errors:
                        self.wait(lane)
                except Exception as e:
                    errors.append(e)
                waited.put(lane)

        def drain():
            while True:
                lane = waited.get()
                if lane is None:
                    merged.put(None)
                    return
                got = None
                try:
                    if not errors:
                        got = self.finish_collect(lane)
                except Exception as e:                             # surfaces in the submitting thread
                    errors.append(e)
                    self.lanes[lane].busy = False
                merged.put((lane, got))

        def hand_over():
            while True:
                item = merged.get()
                if item is None:
                    return
                lane, got = item
                try:
                    if got is not None and not errors:
                        self.current_lane = lane                   # (consume may hand the lane's buffers to write_files)
                        t0 = time.perf_counter()
                        consume(*got)
                        self.times["consume"] += time.perf_counter() - t0
                except Exception as e:
                    errors.append(e)
                free.put(lane)

        th0 = threading.Thread(target=waiter, daemon=True)
        th = threading.Thread(target=drain, daemon=True)
        th2 = threading.Thread(target=hand_over, daemon=True)
        th0.start()
        th.start()
        th2.start()
        try:
            for scenes, inserts, min_points, tag in batches:
                if errors:
                    break
                t0 = time.perf_counter()
                lane = free.get()
                self.times["wait_free_lane"] += time.perf_counter() - t0
                if errors:
                    break
                if isinstance(scenes, tuple) and len(scenes) == 3 and scenes[0] == "files":
                    self.submit_files(lane, scenes[1], scenes[2], inserts, min_points, tag)
                else:
                    self.submit(lane, scenes, inserts, min_points, tag)
                submitted.put(lane)
        finally:
            submitted.put(None)
            th0.join()
            th.join()
            th2.join()
        if errors:
            # lanes that were submitted but never collected still have work on the device: wait for it before the caller
            # lets go of their buffers (kernels writing into memory the allocator has handed to somebody else end in a
            # memory fault, as a soak showed after a frame had been flagged)
            for ln in self.lanes:
                try:
                    ln.stream.synchronize()
                except Exception:
                    pass
                ln.busy = False
            raise errors[0]
