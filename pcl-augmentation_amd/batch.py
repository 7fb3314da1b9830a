"""Batched, HBM-resident driver of the insert loop (SS Real3DAug/insertion.py:371-381, :449-545).

``SceneBatch`` owns the device arrays of ``r3d_batch_t`` (as PyTorch-ROCm tensors) for B
independent scenes and advances all of them in lock step: ``begin`` = the scene field of view of
the first insert, ``insert`` = one placement candidate per scene, ``finish`` = the merged clouds in
the byte layout of ``velodyne/*.bin`` / ``labels/*.label`` / ``check/*.bin``.  Scenes are
independent, so several GPUs simply take disjoint scene shards (``shard_indices``); there is no
collective on the data path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def shard_indices(n_scenes: int, rank: int, world_size: int):
    """Scene i -> GPU i mod G: the indices rank ``rank`` processes."""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside [0, world_size)")
    return list(range(rank, n_scenes, world_size))


class SceneBatch:
    # diagnostics of the most recent augment_batch.  The class attributes are what a single-threaded caller (the tests)
    # reads; augment_batch also leaves both on the batch INSTANCE it worked on, which is what holds when several threads run
    # augment_batch at a time (AugmentPipeline.run(lanes > 1): every worker thread has its own batches)
    last_rebases = 0      # rebases counted
    edge_risk_total = (0, 0)  # points within 1e-12 of a bin edge the batch's kernels have met (scene, sample)
    last_level1 = []      # scenes run once more through the Level-1 kernels (_lib.S_REDO_LEVEL1)

    def __init__(self, B, cap, log_cap, rows=_lib.NUMROW, cols=_lib.NUMCOLUMN, device="cuda:0",
                 exact_projection=False, debug=0, order="auto"):
        """order: the point order of the clouds this batch will see -- "file" (ring-major, firing sequences: any order in
        which consecutive points are neighbours in the range image; nothing is looked at), "any" (every ``begin`` looks at
        the chunk boxes it has built and numbers the points of an unordered cloud anew, internally: four small launches),
        "auto" (default): as "any" until a batch has come through without an unordered scene, then as "file", with a
        look every 64th ``begin`` -- datasets do not change their order from frame to frame.  The results never depend on
        it; a wrong "file" costs time (every insert then walks the whole cloud)."""
        torch = _lib.require_gpu()
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device(device)
        if order not in ("auto", "file", "any"):
            raise ValueError("order: 'auto', 'file' or 'any'")
        self.order, self._begins, self._looked = order, 0, False
        with _lib.on(self.device):
            self._init(B, cap, log_cap, rows, cols, exact_projection, debug)

    def _init(self, B, cap, log_cap, rows, cols, exact_projection, debug):
        torch = self.torch
        cap = (int(cap) + 63) // 64 * 64          # whole 64-point chunks per slab (chunk tables of the placement search)
        self.B, self.cap, self.log_cap, self.rows, self.cols = int(B), int(cap), int(log_cap), int(rows), int(cols)
        self.placed_staging = {}                   # PlacedInserter's staging buffers (pinned once per batch object, not per inserter)
        dev = self.device

        def z(shape, dt):
            return torch.zeros(shape, dtype=dt, device=dev)

        self.xyzi = z((B, cap, 4), torch.float32)
        self.label = z((B, cap), torch.int32)            # uint32 bit patterns
        self.pix = z((B, cap), torch.int32)
        self.n_head = z((B,), torch.int32)
        self.n_total = z((B,), torch.int32)
        self.tail_ref = z((B, log_cap), torch.int32)
        self.log5 = z((B, log_cap, 5), torch.float64)
        self.log_birth = z((B, log_cap), torch.int32)
        self.n_log = z((B,), torch.int32)
        self.bounds = z((B, 2), torch.float64)
        self.far_pix = z((B, _lib.FAR_CAP), torch.int32)
        self.n_far = z((B,), torch.int32)
        self.rebase = z((B,), torch.int32)
        self.status = z((B,), torch.int32)
        self.out_xyzi = z((B, cap, 4), torch.float32)
        self.out_label = z((B, cap), torch.int32)
        self.n_out = z((B,), torch.int32)
        self.n_points = z((B,), torch.int32)
        self.n_visible = z((B,), torch.int32)
        self.accepted = z((B,), torch.int32)
        self.check = None
        self.step = 0

        d = _lib.BatchDesc()
        # reserved bit 0: evaluate the reference's float64 formula for every point instead of the
        # verified float32 guess (diagnostic; results are identical, tests/test_gpu_batch.py)
        # `debug`: further diagnostic bits (2 / 4 / 8 / 16 / 32 / 128: force the insert kernel's other routes; 64: verify every
        # speculative evaluation, csrc/r3d_insert.hip)
        import os
        # (diagnostics: the same bits for every batch of the process -- read HERE, by the Python mirror; the library itself
        # reads no environment variable)
        debug = int(debug) | int(os.environ.get("R3D_DEBUG_BITS", "0"))
        d.B, d.rows, d.cols, d.reserved, d.cap, d.log_cap = B, rows, cols, (1 if exact_projection else 0) | int(debug), cap, log_cap
        for name in ("xyzi", "label", "pix", "n_head", "n_total", "tail_ref", "log5", "log_birth", "n_log",
                     "bounds", "far_pix", "n_far", "rebase",
                     "status", "out_xyzi", "out_label", "n_out"):
            setattr(d, name, getattr(self, name).data_ptr())
        d.workspace, d.workspace_bytes = 0, 0
        ws_bytes = self.lib.r3d_batch_workspace_bytes(C.byref(d))
        if ws_bytes == 0:
            raise _lib.R3DError("r3d_batch_workspace_bytes rejected the batch shape")
        self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), ws_bytes
        self.desc = d
        _lib.check(self.lib.r3d_batch_create(C.byref(d), _lib.stream_ptr()), "r3d_batch_create")

    # -- loading --------------------------------------------------------------------------------
    def _staging(self):
        """Pinned host mirrors of the input and output slabs, allocated on first use."""
        if getattr(self, "_pin", None) is None:
            torch = self.torch
            self._pin = {
                "xyzi": torch.empty((self.B, self.cap, 4), dtype=torch.float32, pin_memory=True),
                "label": torch.empty((self.B, self.cap), dtype=torch.int32, pin_memory=True),
                "n": torch.empty((self.B,), dtype=torch.int32, pin_memory=True),
            }
        return self._pin

    @_lib.on_own_device
    def load(self, scenes):
        """scenes: list of (xyzi float32 [n,4], label uint32 [n]) host arrays, one per scene.
        Only the first n rows of every slab are written and uploaded state beyond them is never
        read (n_points bounds every kernel)."""
        assert len(scenes) == self.B
        hx, hl, hn = self.staging_views()
        # the native packer (threads; a Python loop of 256 slab assignments costs 40 ms per batch)
        for s, (x, _) in enumerate(scenes):
            if np.ndim(x) != 2 or np.shape(x)[1] != 4:
                raise ValueError(f"scene {s}: xyzi must have shape [n, 4], got {np.shape(x)}")
        xs = [np.ascontiguousarray(x, dtype=np.float32) for x, _ in scenes]
        ls = [np.ascontiguousarray(l).astype(np.uint32, copy=False).reshape(-1) for _, l in scenes]
        for s, (x, l) in enumerate(zip(xs, ls)):
            if len(x) > self.cap:
                raise ValueError(f"scene {s}: {len(x)} points exceed capacity {self.cap}")
            if len(l) != len(x):
                raise ValueError(f"scene {s}: {len(l)} labels for {len(x)} points")
        hn[:] = [len(x) for x in xs]
        B = self.B
        px = (C.c_void_p * B)(*[x.ctypes.data for x in xs])
        pl = (C.c_void_p * B)(*[l.ctypes.data for l in ls])
        _lib.check(self.lib.r3d_host_pack_frames(px, pl, hn.ctypes.data, B, self.cap, hx.ctypes.data, hl.ctypes.data, -1, 16),
                   "r3d_host_pack_frames")
        self.upload_staging()

    @_lib.on_own_device
    def upload_staging(self):
        """Upload what was written into ``_staging()`` (pinned views, e.g. by a reader thread)."""
        pin = self._staging()
        self.xyzi.copy_(pin["xyzi"], non_blocking=True)
        self.label.copy_(pin["label"], non_blocking=True)
        self.n_points.copy_(pin["n"], non_blocking=True)
        self._loaded_from_staging = True

    def staging_views(self):
        """NumPy views of the pinned input slabs: xyzi [B,cap,4] float32, label [B,cap] uint32,
        n [B] int32."""
        pin = self._staging()
        return pin["xyzi"].numpy(), pin["label"].numpy().view(np.uint32), pin["n"].numpy()

    @_lib.on_own_device
    def download_views(self):
        """Results as views of pinned host memory (valid until the next download): (xyzi [B,cap,4],
        label [B,cap] uint32, check [B,log_cap,cols] or None, n_out [B], n_log [B])."""
        torch = self.torch
        self.raise_on_status()
        if getattr(self, "_pin_out", None) is None:
            self._pin_out = (torch.empty((self.B, self.cap, 4), dtype=torch.float32, pin_memory=True),
                             torch.empty((self.B, self.cap), dtype=torch.int32, pin_memory=True))
        px, pl = self._pin_out
        px.copy_(self.out_xyzi, non_blocking=True)
        pl.copy_(self.out_label, non_blocking=True)
        ck = None
        if self.check is not None:
            if getattr(self, "_pin_ck", None) is None or self._pin_ck.shape != self.check.shape:
                self._pin_ck = torch.empty(self.check.shape, dtype=torch.float32, pin_memory=True)
            self._pin_ck.copy_(self.check, non_blocking=True)
            ck = self._pin_ck.numpy()
        n_out = self.n_out.cpu().numpy()
        n_log = self.n_log.cpu().numpy()
        torch.cuda.current_stream().synchronize()
        return px.numpy(), pl.numpy().view(np.uint32), ck, n_out, n_log

    @_lib.on_own_device
    def download_delta_views(self, check_cols=5, threads=16):
        """``finish`` + ``download_views`` for a batch whose frames the host still holds (``load``: the pinned staging):
        only the DELTA comes back from the device -- one alive bit per point, the inserted points, counters: 0.1 MB per
        frame instead of 2.5 MB -- and the merged clouds, labels and check rows are put together on the host
        (``r3d_host_merge_frames``: the bytes ``r3d_batch_finish`` would have written).  No compaction runs on the device.
        Returns what ``download_views`` returns; the views are valid until the next call."""
        torch = self.torch
        if not getattr(self, "_loaded_from_staging", False):
            self.finish(check_cols)
            return self.download_views()
        B, cap, log_cap, chunks = self.B, self.cap, self.log_cap, (self.cap + 63) // 64
        cc = max(int(check_cols), 4)
        d = getattr(self, "_delta", None)
        if d is None or d["cc"] != cc:
            pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
            d = self._delta = {
                "cc": cc,
                "d_alive": z((B, chunks), torch.int64), "d_tail_xyzi": z((B, log_cap, 4), torch.float32),
                "d_tail_label": z((B, log_cap), torch.int32), "d_counts": z((2, B), torch.int32),
                "h_alive": pin((B, chunks), torch.int64), "h_tail_xyzi": pin((B, log_cap, 4), torch.float32),
                "h_tail_label": pin((B, log_cap), torch.int32), "h_counts": pin((2, B), torch.int32),
                "h_status": pin((B,), torch.int32), "h_n_log": pin((B,), torch.int32),
                "out_xyzi": torch.empty((B, cap, 4), dtype=torch.float32), "out_label": torch.empty((B, cap), dtype=torch.int32),
                "out_check": torch.empty((B, log_cap, cc), dtype=torch.float32), "n_out": torch.zeros((B,), dtype=torch.int32),
            }
        _lib.check(self.lib.r3d_batch_export_delta(C.byref(self.desc), d["d_alive"].data_ptr(), d["d_tail_xyzi"].data_ptr(),
                                                   d["d_tail_label"].data_ptr(), log_cap, d["d_counts"].data_ptr(), _lib.stream_ptr()),
                   "r3d_batch_export_delta")
        for h, dv in (("h_alive", "d_alive"), ("h_tail_xyzi", "d_tail_xyzi"), ("h_tail_label", "d_tail_label"), ("h_counts", "d_counts")):
            d[h].copy_(d[dv], non_blocking=True)
        d["h_status"].copy_(self.status, non_blocking=True)
        d["h_n_log"].copy_(self.n_log, non_blocking=True)
        order = self.point_order_device().cpu() if self._looked else None
        torch.cuda.current_stream().synchronize()
        if order is not None:
            self.note_point_order(order.numpy())
        st = d["h_status"].numpy()
        for s in np.nonzero(st)[0]:
            _lib.raise_status(int(st[s]), f"scene {s}")
        pin_in = self._staging()
        _lib.check(self.lib.r3d_host_merge_frames(
            pin_in["xyzi"].data_ptr(), pin_in["label"].data_ptr(), cap, d["h_alive"].data_ptr(), chunks, d["h_tail_xyzi"].data_ptr(),
            d["h_tail_label"].data_ptr(), log_cap, d["h_counts"].data_ptr(), B, d["out_xyzi"].data_ptr(), d["out_label"].data_ptr(), cap,
            d["n_out"].data_ptr(), d["out_check"].data_ptr() if check_cols else None, log_cap, cc, int(threads)), "r3d_host_merge_frames")
        ck = d["out_check"].numpy()[:, :, :check_cols] if check_cols else None
        return d["out_xyzi"].numpy(), d["out_label"].numpy().view(np.uint32), ck, d["n_out"].numpy(), d["h_n_log"].numpy()

    @_lib.on_own_device
    def load_device(self, xyzi, label, n_points):
        """Same from tensors already on the device (copied into the batch slabs)."""
        self.xyzi.copy_(xyzi)
        self.label.copy_(label)
        self.n_points.copy_(n_points)
        self._loaded_from_staging = False

    # -- the three phases -----------------------------------------------------------------------
    def _order_bit(self):
        """Sets / clears R3D_B_FILE_ORDER in the descriptor for the begin that follows (see ``order`` of the constructor)."""
        if self.order == "file":
            look = False
        elif self.order == "any":
            look = True
        else:
            # what the last look found arrives with the next status read-back (note_point_order); until one has: look
            look = not getattr(self, "_file_order_seen", False) or self._begins % 64 == 0
        self._begins += 1
        self._looked = look
        if look:
            self.desc.reserved &= ~_lib.B_FILE_ORDER
        else:
            self.desc.reserved |= _lib.B_FILE_ORDER

    def note_point_order(self, n_virtual):
        """What the last ``begin`` that looked has found, read back by whoever synchronises anyway (``raise_on_status``, the
        streamed lanes' collect): n_virtual[s] > 0 = scene s was numbered anew."""
        if self._looked:
            self._file_order_seen = not bool(np.any(np.asarray(n_virtual) > 0))

    @_lib.on_own_device
    def point_order_device(self):
        """Device tensor [B] int32: the points of scene s the last ``begin`` numbered anew (0: slab order kept)."""
        if getattr(self, "_n_virtual", None) is None:
            self._n_virtual = self.torch.zeros((self.B,), dtype=self.torch.int32, device=self.device)
        _lib.check(self.lib.r3d_batch_point_order(C.byref(self.desc), C.c_void_p(self._n_virtual.data_ptr()), _lib.stream_ptr()),
                   "r3d_batch_point_order")
        return self._n_virtual

    @_lib.on_own_device
    def begin(self):
        self.step = 0
        self._xyz_only = False
        self._rebegin = self.begin                      # how run_inserts' time-out fallback starts the batch again
        self._order_bit()
        _lib.check(self.lib.r3d_batch_begin(C.byref(self.desc), C.c_void_p(self.n_points.data_ptr()),
                                            _lib.stream_ptr()), "r3d_batch_begin")

    @_lib.on_own_device
    def begin_xyz(self, xyz3):
        """Step 0 from x y z alone: ``xyz3`` = device tensor [B, cap, 3] float32 (12 bytes per point over the link instead of
        the 16 of a velodyne row), ``n_points`` already set.  For delta mode only -- the frames' intensities and labels stay on
        the host, ``finish`` / ``download_views`` refuse (``r3d_batch_begin_xyz``)."""
        self.step = 0
        self._rebegin = lambda: self.begin_xyz(xyz3)
        self._xyz_only = True
        self._order_bit()
        _lib.check(self.lib.r3d_batch_begin_xyz(C.byref(self.desc), C.c_void_p(xyz3.data_ptr()), C.c_void_p(self.n_points.data_ptr()),
                                                _lib.stream_ptr()), "r3d_batch_begin_xyz")

    @_lib.on_own_device
    def begin_f64(self, scenes5):
        """Step 0 for clouds with genuine float64 coordinates (the Waymo flavour, SS tools/datasets.py:240-262):
        scenes5[s] = N x 5 float64 rows [x y z intensity label].  The frame's points are kept in the log next
        to the inserted ones (``log_cap`` >= N + inserted points), every later step works on the float64 values;
        read the result with ``results_f64``."""
        torch = self.torch
        assert len(scenes5) == self.B
        if getattr(self, "_rows5", None) is None:
            self._rows5 = torch.zeros((self.B, self.cap, 5), dtype=torch.float64, device=self.device)
            self._rows5_pin = torch.zeros((self.B, self.cap, 5), dtype=torch.float64, pin_memory=True)
        host, n = self._rows5_pin.numpy(), np.zeros(self.B, dtype=np.int32)
        for s, rows in enumerate(scenes5):
            rows = np.asarray(rows, dtype=np.float64)
            if len(rows) > self.cap or len(rows) > self.log_cap:
                raise ValueError(f"scene {s}: {len(rows)} points exceed capacity {min(self.cap, self.log_cap)}")
            n[s] = len(rows)
            host[s, :len(rows)] = rows[:, :5]
        self._rows5.copy_(self._rows5_pin, non_blocking=True)
        self.n_frame = n.copy()
        self._begin_rows5()

    @_lib.on_own_device
    def _begin_rows5(self):
        """Step 0 from the float64 rows already on the device (``begin_f64``, and again from the time-out fallback of
        ``run_inserts``: the rows and counts are untouched by the inserts)."""
        self.step = 0
        self._xyz_only = False
        self._rebegin = self._begin_rows5
        self._order_bit()
        self.n_points.copy_(self.torch.from_numpy(self.n_frame))
        _lib.check(self.lib.r3d_batch_begin_f64(C.byref(self.desc), C.c_void_p(self._rows5.data_ptr()),
                                                C.c_void_p(self.n_points.data_ptr()), _lib.stream_ptr()),
                   "r3d_batch_begin_f64")

    @_lib.on_own_device
    def results_f64(self):
        """After ``begin_f64`` and the inserts: per scene (merged N' x 5 float64 rows [x y z intensity label] in
        the reference's order -- surviving frame points, then surviving inserted points --, added M x 5 float64 =
        all_visible_parts).  Coordinates and labels come from ``r3d_batch_export_rows`` (exact float64), the
        intensity column of the MERGED rows from the compaction (``finish``), i.e. rounded to float32 -- exact for every
        dataset flavour of the reference (their intensities are float32 in the files) and what ``save_data`` stores anyway;
        the ADDED rows come from the log and keep the float64 value they were given."""
        rows4, n_rows = self.export_rows()
        self.finish(check_cols=0)
        rows4, n_rows = rows4.cpu().numpy(), n_rows.cpu().numpy()
        inten, n_out = self.out_xyzi[:, :, 3].cpu().numpy(), self.n_out.cpu().numpy()
        log5, n_log = self.log5.cpu().numpy(), self.n_log.cpu().numpy()
        out = []
        for s in range(self.B):
            assert n_rows[s] == n_out[s]
            m = np.empty((n_rows[s], 5))
            m[:, 0:3], m[:, 3], m[:, 4] = rows4[s, :n_rows[s], 0:3], inten[s, :n_out[s]], rows4[s, :n_rows[s], 3]
            out.append((m, log5[s, self.n_frame[s]:n_log[s]].copy()))
        return out

    @_lib.on_own_device
    def insert_device(self, samples5, sample_off, min_points, active=None, new_slot=True):
        """One candidate per scene from device tensors; returns (n_visible, accepted) tensors.

        ``new_slot`` starts a new insert slot (a new step number); further candidates of the same
        slot pass ``new_slot=False`` together with ``active`` = scenes still without an accept.
        """
        if new_slot:
            self.step += 1
        _lib.check(self.lib.r3d_batch_insert(
            C.byref(self.desc), C.c_void_p(samples5.data_ptr()), C.c_void_p(sample_off.data_ptr()),
            C.c_void_p(min_points.data_ptr()), C.c_void_p(active.data_ptr() if active is not None else 0),
            self.step, C.c_void_p(self.n_visible.data_ptr()), C.c_void_p(self.accepted.data_ptr()),
            _lib.stream_ptr()), "r3d_batch_insert")
        return self.n_visible, self.accepted

    @_lib.on_own_device
    def insert_first_device(self, cand, cand_stride, sample_off, min_points, active, n_possible, first_cand, n_cand, accepted_at,
                            replay_last=False, new_slot=True):
        """The candidate loop of one insert slot in one call (``r3d_batch_insert_first``): candidate j of all scenes is the
        packed sample list at ``cand[j * cand_stride:]`` (float64 tensor, what the placement search writes); per scene the
        candidates ``first_cand + j < n_possible[s]`` are tried in order until one is accepted -- ``accepted_at[s]`` (int32
        device tensor, -1 where nothing has been accepted yet) then holds its number.  Returns the n_visible tensor."""
        if new_slot:
            self.step += 1
        _lib.check(self.lib.r3d_batch_insert_first(
            C.byref(self.desc), C.c_void_p(cand.data_ptr()), int(cand_stride), C.c_void_p(sample_off.data_ptr()),
            C.c_void_p(min_points.data_ptr()), C.c_void_p(active.data_ptr() if active is not None else 0),
            C.c_void_p(n_possible.data_ptr()), int(first_cand), int(n_cand), self.step, 1 if replay_last else 0,
            C.c_void_p(self.n_visible.data_ptr()), C.c_void_p(accepted_at.data_ptr()), _lib.stream_ptr()), "r3d_batch_insert_first")
        return self.n_visible

    @_lib.on_own_device
    def insert_many_device(self, packed, min_points):
        """Several slots with one candidate each in ONE launch (``r3d_batch_insert_many``): packed =
        list of (samples5, sample_off) device tensors per slot, min_points = list of int32 device
        tensors.  Returns (n_visible [K,B], accepted [K,B]) device tensors."""
        torch = self.torch
        K = len(packed)
        if getattr(self, "_many_out", None) is None or self._many_out[0].shape[0] < K:
            self._many_out = (torch.zeros((max(K, 8), self.B), dtype=torch.int32, device=self.device),
                              torch.zeros((max(K, 8), self.B), dtype=torch.int32, device=self.device))
        nv, acc = self._many_out[0][:K], self._many_out[1][:K]      # every slot of every scene is written by the call
        ptrs = lambda xs: (C.c_void_p * K)(*[C.c_void_p(x) for x in xs])
        _lib.check(self.lib.r3d_batch_insert_many(
            C.byref(self.desc), K, ptrs([p[0].data_ptr() for p in packed]), ptrs([p[1].data_ptr() for p in packed]),
            ptrs([m.data_ptr() for m in min_points]), None, self.step + 1, ptrs([nv[k].data_ptr() for k in range(K)]),
            ptrs([acc[k].data_ptr() for k in range(K)]), _lib.stream_ptr()), "r3d_batch_insert_many")
        self.step += K
        return nv, acc

    @_lib.on_own_device
    def pack_samples(self, samples):
        """list of (M x 5 float64 | None) per scene -> (samples5, sample_off) device tensors."""
        torch = self.torch
        assert len(samples) == self.B
        off = np.zeros(self.B + 1, dtype=np.int64)
        np.cumsum(np.fromiter((0 if smp is None else len(smp) for smp in samples), dtype=np.int64, count=self.B), out=off[1:])
        parts = [smp for smp in samples if smp is not None and len(smp)]
        # (one concatenation: a Python loop of 256 slice assignments cost 0.4 ms of a placed slot's 4.5)
        rows = np.concatenate(parts, axis=0).astype(np.float64, copy=False) if parts else np.zeros((1, 5), dtype=np.float64)
        return (torch.from_numpy(np.ascontiguousarray(rows)).to(self.device), torch.from_numpy(off).to(self.device))

    @_lib.on_own_device
    def insert(self, samples, min_points, active=None, new_slot=True):
        """One candidate per scene (insertion.py:453-526): (n_visible, accepted) as arrays.  ``min_points[s] < 0`` replays a
        candidate that HAS been rejected the way the reference's driver is left with it -- the scene without the points the
        candidate covers, without the candidate (:468-471, not :526): the scene is not changed, ``export_rows`` returns that
        copy until the scene's next candidate, ``adopt_rejected`` makes it the scene."""
        torch = self.torch
        s5, off = self.pack_samples(samples)
        mp = torch.from_numpy(np.asarray(min_points, dtype=np.int32)).to(self.device)
        act = None if active is None else torch.from_numpy(np.asarray(active, dtype=np.int32)).to(self.device)
        nv, acc = self.insert_device(s5, off, mp, act, new_slot)
        self._keep = (s5, off, mp, act)          # keep the inputs alive until the stream has run
        return nv.cpu().numpy().copy(), acc.cpu().numpy().copy()

    @_lib.on_own_device
    def export_rows(self):
        """The current merged clouds as float64 rows [x y z label] (what the placement search reads,
        insertion.py:433): (rows [B,cap,4], n_rows [B]) device tensors; the batch is left unchanged."""
        torch = self.torch
        if getattr(self, "rows4", None) is None:
            self.rows4 = torch.empty((self.B, self.cap, 4), dtype=torch.float64, device=self.device)
            self.n_rows = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.r3d_batch_export_rows(C.byref(self.desc), C.c_void_p(self.rows4.data_ptr()),
                                                  C.c_void_p(self.n_rows.data_ptr()), _lib.stream_ptr()),
                   "r3d_batch_export_rows")
        return self.rows4, self.n_rows

    @_lib.on_own_device
    def export_alive(self):
        """The alive word of every 64-point chunk of every scene, slab order ([B, chunks] int64 device tensor): the living
        points of the cloud ``export_rows`` shows, without the rows (``r3d_batch_export_alive``)."""
        torch = self.torch
        if getattr(self, "_alive_words", None) is None:
            self._alive_words = torch.zeros((self.B, (self.cap + 63) // 64), dtype=torch.int64, device=self.device)
        _lib.check(self.lib.r3d_batch_export_alive(C.byref(self.desc), C.c_void_p(self._alive_words.data_ptr()), _lib.stream_ptr()),
                   "r3d_batch_export_alive")
        return self._alive_words

    @_lib.on_own_device
    def adopt_rejected(self, active=None):
        """The copy a rejected candidate has left (``insert`` with ``min_points < 0``) becomes the scene, where there is one
        (and ``active[s]``): what the reference's driver goes on with when no further candidate restores the backup
        (insertion.py:453, then :373 or save_data)."""
        torch = self.torch
        act = None
        if active is not None:
            act = active if hasattr(active, "data_ptr") else torch.from_numpy(np.asarray(active, dtype=np.int32)).to(self.device)
        _lib.check(self.lib.r3d_batch_adopt_rejected(C.byref(self.desc), C.c_void_p(act.data_ptr() if act is not None else 0),
                                                     _lib.stream_ptr()), "r3d_batch_adopt_rejected")
        self._keep_adopt = act

    @_lib.on_own_device
    def finish(self, check_cols=5):
        torch = self.torch
        if getattr(self, "_xyz_only", False):
            raise _lib.R3DError("finish after begin_xyz: the frames' intensities and labels are not on the device (take the delta)")
        if check_cols:
            # rows beyond n_log[s] are never read: the buffer is kept between calls
            if getattr(self, "_check_buf", None) is None or self._check_buf.shape[2] != check_cols:
                self._check_buf = torch.zeros((self.B, self.log_cap, check_cols), dtype=torch.float32, device=self.device)
            self.check = self._check_buf
            cp = C.c_void_p(self.check.data_ptr())
        else:
            self.check, cp = None, C.c_void_p(0)
        _lib.check(self.lib.r3d_batch_finish(C.byref(self.desc), cp, int(check_cols or 5), _lib.stream_ptr()),
                   "r3d_batch_finish")

    def count_pairs(self, on=True):
        """Descriptor bit 4096: the insert kernels count per pair (pairs, chunks listed, parked, committed from a record) --
        atomics of every pair on a few addresses, hence off unless somebody reads them (``debug_counters``)."""
        if on:
            self.desc.reserved |= 4096
        else:
            self.desc.reserved &= ~4096

    @_lib.on_own_device
    def debug_counters(self, reset=True):
        """The insert kernels' diagnostic counters (r3d_batch_debug_counters) as a dict."""
        out = (C.c_int32 * 64)()
        _lib.check(self.lib.r3d_batch_debug_counters(C.byref(self.desc), out, (1 if reset else 0) | 2 | 4, _lib.stream_ptr()),
                   "r3d_batch_debug_counters")
        names = ["pool_exhausted", "tiles_pooled", "evaluated_twice", "verify_runs", "verify_mismatch", "hits_overflow", "deferred_scenes",
                 "rebases_in_chain", "rebase_for_sample_point_outside_bounds", "rebase_for_culled_holder", "rebase_from_far_pass",
                 "rebase_without_reason"]
        d = dict(zip(names, list(out)))
        if any(out[12:16]):              # a diagnostic build (-DR3D_CHECK) counted index checks that failed: [12 + (code & 3)]
            d["check_failures"] = list(out[12:16])
            d["check_notes"] = list(out[16:32])               # ... and what it noted about the first one (csrc/r3d_insert.hip)
        # round 5 (no workgroup waits for another): pairs committed by the workgroup that evaluated them, pairs left to the
        # workgroup that finishes their predecessor (with / without a record), pairs committed from such a record
        pairs, listed = int(out[32]), int(out[33])
        d.update(pairs_committed_by_their_evaluator=pairs, parked_with_record=int(out[34]), parked_unevaluated=int(out[35]),
                 committed_from_record=int(out[36]), scenes_in_sorted_order=int(out[37]),
                 sparse_tiles=int(out[38]), sparse_tiles_beyond_the_lds=int(out[39]),
                 # points whose pixel the reference formula decided within 1e-12 of a bin edge (scene points at step 0 /
                 # rebase, sample points per evaluation): where an ULP of arctan2 / arccos could move a pixel
                 bin_edge_risk_scene_points=int(out[40]), bin_edge_risk_sample_points=int(out[41]),
                 # scenes begun under R3D_B_FILE_ORDER (order="file", or "auto" between two looks) whose chunk boxes say that
                 # their points come in no file order: every insert of such a scene walks the whole cloud (time, never results)
                 unordered_scenes_under_file_order_promise=int(out[42]),
                 chunks_listed_per_pair=round(listed / pairs, 1) if pairs else None)
        return d

    @_lib.on_own_device
    def pixel_ids(self):
        """The pixel id of every point, in the order of the slabs, as the reference numbers it (row * cols + col,
        insertion.py:116): the batch keeps (row << 16) | column, for a scene in virtual order under another point numbering
        (``r3d_batch_export_pix``)."""
        out = self.torch.zeros((self.B, self.cap), dtype=self.torch.int32, device=self.device)
        _lib.check(self.lib.r3d_batch_export_pix(C.byref(self.desc), C.c_void_p(out.data_ptr()), _lib.stream_ptr()), "r3d_batch_export_pix")
        return out.cpu().numpy()

    # -- results --------------------------------------------------------------------------------
    @_lib.on_own_device
    def raise_on_status(self):
        if self._looked:
            nv = self.point_order_device()
        st = self.status.cpu().numpy()
        if self._looked:
            self.note_point_order(nv.cpu().numpy())
        for s in np.nonzero(st)[0]:
            _lib.raise_status(int(st[s]), f"scene {s}")

    @_lib.on_own_device
    def results(self, delta_check_cols=None):
        """Per scene: (xyzi float32 [n,4], label uint32 [n], check float32 [m,cols]), copies.  After ``finish``; or, with
        ``delta_check_cols`` (4 / 5 / 0) INSTEAD of ``finish``: by the delta and the host merge (``download_delta_views``)."""
        ox, ol, ck, n_out, n_log = self.download_views() if delta_check_cols is None else self.download_delta_views(delta_check_cols)
        one = lambda s: (ox[s, :n_out[s]].copy(), ol[s, :n_out[s]].copy(), ck[s, :n_log[s]].copy() if ck is not None else None)
        if self.B < 16:
            return [one(s) for s in range(self.B)]
        from concurrent.futures import ThreadPoolExecutor      # NumPy copies release the interpreter lock
        with ThreadPoolExecutor(max_workers=8) as pool:
            return list(pool.map(one, range(self.B)))

    @_lib.on_own_device
    def run_inserts(self, candidates, min_points):
        """The candidate loop of insertion.py:449-545 for every scene, after ``begin``: candidates
        of a slot are tried in order, the slot's "still open" mask lives on the device, and the
        host synchronises once, at the end.  Returns accepted[s][k] = index of the accepted
        candidate of insert k of scene s, or -1."""
        torch = self.torch
        B = self.B
        k_max = max(len(c) for c in candidates)
        if k_max and all(len(slot) <= 1 for c in candidates for slot in c):
            # one candidate per slot everywhere: all slots in one call (r3d_batch_insert_many)
            packed, needs = [], []
            for k in range(k_max):
                packed.append(self.pack_samples([c[k][0] if k < len(c) and len(c[k]) else None for c in candidates]))
                needs.append(torch.from_numpy(np.asarray([min_points[s][k] if k < len(candidates[s]) else 0
                                                          for s in range(B)], dtype=np.int32)).to(self.device))
            _, acc = self.insert_many_device(packed, needs)
            self._keep = (packed, needs)
            acc_h = acc.cpu().numpy()                                        # the one synchronisation
            bad = int((self.status & _lib.S_CHAIN_TIMEOUT).sum().item())
            if bad and self.step == k_max:
                # a slot gave up waiting for its scene's previous slot (the chain kernel leans on the order in which
                # workgroups are dispatched): do the batch again right here, one launch per slot -- the frames are
                # still in the slabs, step 0 restores everything the inserts changed
                self.chain_timeouts = getattr(self, "chain_timeouts", 0) + bad
                # (a batch begun with begin_f64 is begun that way again: its frame lives in the float64 rows and the log)
                getattr(self, "_rebegin", self.begin)()
                accs = [self.insert_device(p[0], p[1], nd)[1].clone() for p, nd in zip(packed, needs)]
                acc_h = torch.stack(accs).cpu().numpy()
                bad = int((self.status & _lib.S_CHAIN_TIMEOUT).sum().item())
            if bad:
                raise _lib.R3DError("r3d_batch_insert_many could not order the slots of a scene on this device "
                                    "(status R3D_S_CHAIN_TIMEOUT); SceneBatch(debug=_lib.B_SLOT_LAUNCHES) inserts slot by slot")
            return [[0 if acc_h[k, s] else -1 for k in range(len(candidates[s]))] for s in range(B)]
        log, keep = [], []
        for k in range(k_max):
            n_cand = max(len(c[k]) if k < len(c) else 0 for c in candidates)
            need = torch.from_numpy(np.asarray([min_points[s][k] if k < len(candidates[s]) else 0
                                                for s in range(B)], dtype=np.int32)).to(self.device)
            still_open = None
            for ci in range(n_cand):
                smp = [candidates[s][k][ci] if k < len(candidates[s]) and ci < len(candidates[s][k]) else None
                       for s in range(B)]
                s5, off = self.pack_samples(smp)
                _, acc = self.insert_device(s5, off, need, still_open, new_slot=(ci == 0))
                acc = acc.clone()
                log.append((k, ci, acc))
                keep.append((s5, off, need, still_open))
                if ci + 1 < n_cand:
                    still_open = (1 - acc) if still_open is None else still_open * (1 - acc)
        accepted = [[-1] * len(c) for c in candidates]
        if log:
            flags = torch.stack([a for _, _, a in log]).cpu().numpy()      # the one synchronisation
            for (k, ci, _), row in zip(log, flags):
                for s in np.nonzero(row)[0]:
                    if k < len(accepted[s]) and accepted[s][k] < 0:
                        accepted[s][k] = ci
        return accepted


def level2_takes(rows, cols):
    """Does the batched path (Level 2) take this range-image shape?  (check_batch of csrc/r3d_batch.hip.)"""
    rows, cols = int(rows), int(cols)
    return (0 < rows <= 65535 and 0 < cols <= 65535 and cols % 32 == 0 and rows * cols < (1 << 24)
            and ((cols + 1) + rows + 2) * 2 * 4 <= 60 * 1024)


def augment_batch(scenes, candidates, min_points, rows=_lib.NUMROW, cols=_lib.NUMCOLUMN, device="cuda:0",
                  check_cols=5, reuse=None, debug=0):
    """Run whole frames through the insert loop on one GPU.

    scenes[s] = (xyzi float32 [n,4], label uint32 [n]); candidates[s][k] = ordered list of M x 5
    float64 placement candidates tried for insert k of scene s (the first one whose visible part
    reaches min_points[s][k] points is merged, insertion.py:449-526).  Returns
    (results, accepted) with results as ``SceneBatch.results`` and accepted[s][k] = index of the
    accepted candidate or -1.  ``reuse``: optional dict in which the ``SceneBatch`` is kept between
    calls (keyed by B) instead of being allocated every time.
    """
    B = len(scenes)
    if not level2_takes(rows, cols):
        # a range image the batched kernels are not built for (columns that are no multiple of 32: the bit images are rows of
        # 32-pixel words; 2^24 pixels and more; edge tables beyond the projection kernel's LDS): every frame through the
        # Level-1 kernels, which have none of these limits -- the reference takes any NUMROW / NUMCOLUMN (insertion.py:22-23)
        from . import level1
        out = [level1.augment_scene(x, l, candidates[s], min_points[s], rows, cols, device, check_cols) for s, (x, l) in enumerate(scenes)]
        SceneBatch.last_rebases, SceneBatch.last_level1 = 0, list(range(B))
        return [o[0] for o in out], [o[1] for o in out]
    grow = max(sum(max((len(x) for x in slot), default=0) for slot in c) for c in candidates)
    cap = max(len(x) for x, _ in scenes) + grow
    batch = reuse.get(B) if reuse is not None else None
    if batch is None or batch.cap < cap or batch.log_cap < max(grow, 1) or (batch.rows, batch.cols) != (rows, cols):
        # a descriptor is re-usable for any later call that fits (a rebase rewrites only the slabs
        # that load() overwrites anyway); `reuse` is the caller's {B: SceneBatch} cache
        batch = SceneBatch(B, int(cap * 1.05) + 64 if reuse is not None else cap, max(grow, 1) * (2 if reuse is not None else 1),
                           rows, cols, device, debug=debug)
        if reuse is not None:
            reuse[B] = batch
    batch.load(scenes)
    batch.begin()
    accepted = batch.run_inserts(candidates, min_points)
    batch.finish(check_cols)
    batch.last_rebases = SceneBatch.last_rebases = int(batch.rebase.sum().item())
    # points decided by the reference formula within 1e-12 of a bin edge since this batch object was made (scene, sample):
    # what tools/fuzz_parity.py reports beside its comparisons
    cnt = batch.debug_counters(reset=False)
    batch.edge_risk_total = SceneBatch.edge_risk_total = (cnt["bin_edge_risk_scene_points"], cnt["bin_edge_risk_sample_points"])
    batch.last_level1 = SceneBatch.last_level1 = []
    # A frame beyond the batched kernels' limits -- an insert window that exceeds a CU's LDS (an object a few metres from the
    # sensor on a grid several times the reference's), a sample of more than R3D_MAX_SAMPLE points, more than R3D_FAR_CAP
    # pixels beyond 500 m: once more, alone, through the Level-1 kernels -- whole range images in HBM, no such limit
    # (the reference has none: insertion.py:455-482)
    order = batch.point_order_device() if batch._looked else None
    status = batch.status.cpu().numpy()
    if order is not None:
        batch.note_point_order(order.cpu().numpy())
    redo = [int(s) for s in np.nonzero(status)[0] if _lib.needs_level1(status[s])]
    if redo:
        batch.status[torch_index(batch, redo)] = 0
    results = batch.results()
    if redo:
        from . import level1
        for s in redo:
            results[s], accepted[s] = level1.augment_scene(scenes[s][0], scenes[s][1], candidates[s], min_points[s], rows, cols,
                                                           device, check_cols)
        batch.last_level1 = SceneBatch.last_level1 = redo
    return results, accepted


def torch_index(batch, idx):
    return batch.torch.as_tensor(idx, dtype=batch.torch.int64, device=batch.device)


def run_sharded(n_scenes, process_shard, rank=None, world_size=None, group=None):
    """Scene-sharded execution across the GPUs of a node: rank r processes
    scenes r, r+G, r+2G, ... with ``process_shard(indices) -> list of per-scene results`` and the
    per-rank results are gathered on every rank in scene order.  No collective touches the data
    path; the single ``all_gather_object`` at the end only returns results to the caller (a
    production driver writes its own files instead and skips it with ``group=False``).

    ``process_shard`` is the caller's per-shard work: ``augment_batch`` on the rank's device, as a rule (this function only
    deals the scene indices out and gathers what comes back; it runs nothing itself).
    """
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard_indices(n_scenes, rank, world_size)
    local = process_shard(mine)
    if len(local) != len(mine):
        raise ValueError("process_shard must return one result per scene index")
    if world_size == 1 or group is False:
        return dict(zip(mine, local))
    gathered = [None] * world_size
    dist.all_gather_object(gathered, list(zip(mine, local)), group=group)
    merged = {}
    for part in gathered:
        merged.update(dict(part))
    return merged
