"""Insert loop with placement search for a batch of frames: what the reference's driver does per
frame between insertion.py:380 and :545, with both halves on the GPU.

For one insert slot of every scene: ``find_possible_places`` on the scene's *current* cloud
(find_spot.py:192-273 -> ``r3d_find_possible_places``), then the possible placements are tried
in rotation order and the first one whose visible part reaches ``min_points`` is merged
(insertion.py:449-526 -> ``r3d_batch_insert``); the accepted object's box joins the scene's
annotations (:535).  Which sample is tried for which slot stays with the caller (the reference
shuffles its object database, :396-400).  The candidate clouds never leave HBM: the search writes
candidate j of all scenes as one packed sample list, which is what the insert call reads.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .places import PlaceBatch, chunk_ranges, upload_map


class PlacedInserter:
    def __init__(self, batch, rich_maps, map_moves, poses, scene_boxes, reference_rejected_state=False, scene_slab=True):
        """batch: a SceneBatch after ``begin``.  Per scene: rich map (2D integer codes), its
        ``move`` (first two entries used), the 4 x 4 pose and the annotated boxes (k x 10:
        centre, quaternion xyzw, length, width, height).

        reference_rejected_state: do what the reference's driver does when every candidate of a sample is rejected -- it
        goes on with the scene WITHOUT the points the last rejected candidate covers (insertion.py:468-471 ran, the next
        candidate's restore :453 did not): the next sample's placement search sees that copy, and when the sample was the
        object's last try (``insert_slot(..., last_try=...)``) the copy becomes the scene.  Default: a rejected candidate
        changes nothing (INTEGRATION.md par. 6)."""
        torch = _lib.require_gpu()
        self.batch, self.torch = batch, torch
        self.reference_rejected_state = bool(reference_rejected_state)
        B = batch.B
        assert len(rich_maps) == len(map_moves) == len(poses) == len(scene_boxes) == B
        # (maps of one shape -- the usual case, one map geometry per dataset -- go up as one slab)
        if all(isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.shape == rich_maps[0].shape and m.ndim == 2 for m in rich_maps):
            # (stacked in pinned memory that belongs to the batch: the copy is asynchronous, and a lane that runs batch after
            # batch on one SceneBatch pins its staging once -- pinning and unpinning per batch stalls every lane of the process)
            shape = (B,) + tuple(rich_maps[0].shape)
            pin = self._staging("maps", lambda: torch.empty(shape, dtype=torch.uint8, pin_memory=True),
                                lambda t: tuple(t.shape) == shape)
            np.stack(rich_maps, out=pin.numpy())
            slab = pin.to(batch.device, non_blocking=True)
            self.maps = [slab[s] for s in range(B)]
        else:
            self.maps = [upload_map(m, batch.device) for m in rich_maps]
        self.moves = [(float(np.asarray(m).reshape(-1)[0]), float(np.asarray(m).reshape(-1)[1])) for m in map_moves]
        self.poses = [np.asarray(p, dtype=np.float64)[:2, :4].reshape(8).copy() for p in poses]
        # the annotated boxes of every scene in one array (rows beyond n_boxes[s] unused): an accepted object's box joins its
        # scene's rows (insertion.py:535) without a Python loop over the batch
        bx = [np.asarray(b, dtype=np.float64).reshape(-1, 10) for b in scene_boxes]
        self.n_boxes = np.array([len(b) for b in bx], dtype=np.int64)
        self.boxes_h = np.zeros((B, max(8, int(self.n_boxes.max()) + 8), 10))
        for s in range(B):
            self.boxes_h[s, :len(bx[s])] = bx[s]
        # original_pcl (insertion.py:360): the clouds as loaded
        # (the counts `load` wrote into the pinned staging, when the batch was loaded that way: no wait for the device)
        pin = getattr(batch, "_pin", None)
        n0 = pin["n"].numpy() if pin is not None and getattr(batch, "_loaded_from_staging", False) else batch.n_points.cpu().numpy()
        self.n_orig = [int(v) for v in n0]
        # (64-point chunks of a scene are whole chunks of the slab: SceneBatch rounds its stride up to a multiple of 64)
        assert batch.cap % 64 == 0
        # Round 6: the search reads the CURRENT cloud of a scene where it stands -- the batch's float32 slab, its labels, the
        # alive words, the log for inserted points (R3D_PQ_SCENE_SLAB) -- instead of float64 rows exported per slot
        # (r3d_batch_export_rows + their chunk ranges: 0.66 ms of a slot's 3.8 on 256 frames), and the ORIGINAL cloud from the
        # same slab (R3D_PQ_ORIG_SLAB: its first n_head rows never move or change) instead of a float64 copy per batch (1 GB
        # on 256 frames).  The chunk ranges of the slab: the original cloud's for whole chunks of frame points (a dead point
        # only leaves its chunk's range wider than needed), "always in reach" for the chunks that hold inserted points.
        # Float32 frames only (begin / begin_xyz).
        n_head_h = batch.n_head.cpu().numpy()
        self.slab = bool(scene_slab) and bool(np.array_equal(n_head_h, np.asarray(self.n_orig)))
        if self.slab:
            self.orig_rows = None
            # (a scene's last chunk may take in rows past its end: its range only gets wider, which is safe)
            self.orig_ranges_all = chunk_ranges(batch.xyzi.view(B * batch.cap, 4)).view(B, batch.cap // 64, 2)
            first_open = torch.from_numpy((n_head_h // 64).astype(np.int64)).to(batch.device)
            open_ = torch.arange(batch.cap // 64, device=batch.device)[None, :] >= first_open[:, None]
            sr = self.orig_ranges_all.clone()
            sr[..., 0][open_] = 0.0
            sr[..., 1][open_] = float("inf")
            self.slab_ranges = sr
            self.n_head_arr = n_head_h.astype(np.int64)
            self.n_scene_h = self.n_head_arr.copy()                      # (points of the slab so far: updated with every slot's results)
        else:
            # ... as packed float64 rows [x y z label]
            self.orig_rows = torch.cat([batch.xyzi[:, :, :3].to(torch.float64),
                                        (batch.label.to(torch.int64) & 0xFFFF).to(torch.float64)[:, :, None]], dim=2).contiguous()
            self.orig_ranges_all = chunk_ranges(self.orig_rows.view(B * batch.cap, 4)).view(B, batch.cap // 64, 2)
        # the slot's staging: one pinned buffer up (made on first use, grows), two small pinned buffers down; kept on the batch
        self._down_f = self._staging("down_f", lambda: torch.empty(B * 11, dtype=torch.float64, pin_memory=True))
        self._down_i = self._staging("down_i", lambda: torch.empty(2 * B, dtype=torch.int32, pin_memory=True))
        self._arange = self._staging("arange", lambda: torch.arange(B, dtype=torch.int64, device=batch.device))
        # what the descriptors of a slot are packed from, per scene, as arrays (insert_slot fills all queries at once)
        self.map_ptr = np.array([m.data_ptr() for m in self.maps], dtype=np.uint64)
        self.map_shape = np.array([m.shape for m in self.maps], dtype=np.int32)
        self.move_arr = np.array(self.moves, dtype=np.float64)
        self.pose_arr = np.array(self.poses, dtype=np.float64)
        self.n_orig_arr = np.array(self.n_orig, dtype=np.int64)

    @property
    def boxes(self):
        """Per scene: its annotated boxes so far (k x 10), the accepted objects' included."""
        return [self.boxes_h[s, :self.n_boxes[s]] for s in range(self.batch.B)]

    def insert_slot(self, samples, annos, ok_labels, ok_maps, min_points, chunk=8, flavours=None, last_try=True):
        """samples[s]: M x 5 float64 or None; annos[s]: the sample's box (10 floats) after
        read_label_line; ok_labels[s] / ok_maps[s]: placement labels / map codes of its class;
        flavours[s] (optional): dict with ``flavour`` / ``collide_label`` / ``collide_dz`` for the
        object-detection rules (``find_spot.od.place_query`` builds them).
        last_try (bool, or one per scene; only with ``reference_rejected_state``): this sample is the last one tried for
        its object -- the reference's while-loop then starts over from the scene as the sample has left it (:373).
        Returns (rotation[s] = accepted rotation number or -1, n_possible[s])."""
        out = self._insert_slot(samples, annos, ok_labels, ok_maps, min_points, chunk, flavours)
        if self.reference_rejected_state:
            lt = np.asarray(last_try, dtype=bool)
            act = np.broadcast_to(lt, (self.batch.B,)).astype(np.int32) if lt.ndim == 0 or lt.shape == (self.batch.B,) else None
            assert act is not None, "last_try: a bool or one per scene"
            if act.any():
                self.batch.adopt_rejected(None if act.all() else act)
        return out

    # -- one insert slot --------------------------------------------------------------------------
    # Round 6: what a slot sends to the device -- descriptors, boxes, the samples' rows, offsets, the masks -- is written into
    # ONE pinned host buffer and goes up with one copy; what it reads back -- per query the accepted candidate, its rotation
    # and annotation, the search's status; per scene the batch's status and point count -- is gathered on the device and comes
    # down with two small copies.  (Before: some 25 copies from / to pageable memory per slot, 5.5 MB of them the
    # annotations of all 360 steps of every query, each one a blocking call into the runtime's staging path.)
    def _staging(self, name, make, fits=lambda t: True):
        """A staging buffer of this batch by name: made once per SceneBatch (one PlacedInserter works on a batch at a time)."""
        held = self.batch.placed_staging
        if name not in held or not fits(held[name]):
            held[name] = make()
        return held[name]

    def _room(self, nbytes):
        """(pinned host buffer, its NumPy view, device buffer) of at least nbytes for a slot's upload."""
        torch = self.torch
        size = int(nbytes * 1.25) + (1 << 20)
        pin = self._staging("up_pin", lambda: torch.empty(size, dtype=torch.uint8, pin_memory=True), lambda t: t.numel() >= nbytes)
        dev = self._staging("up_dev", lambda: torch.empty(pin.numel(), dtype=torch.uint8, device=self.batch.device),
                            lambda t: t.numel() >= nbytes)
        return pin, pin.numpy(), dev

    def _insert_slot(self, samples, annos, ok_labels, ok_maps, min_points, chunk, flavours):
        torch, batch = self.torch, self.batch
        B = batch.B
        who = [s for s in range(B) if samples[s] is not None and len(samples[s])]
        rotation, n_poss = [-1] * B, [0] * B
        if not who:
            return rotation, n_poss
        if self.slab:
            rows, alive, all_ranges = None, batch.export_alive(), self.slab_ranges
        else:
            rows, n_rows = batch.export_rows()
            alive, all_ranges = None, chunk_ranges(rows.view(B * batch.cap, 4)).view(B, batch.cap // 64, 2)
            self.n_scene_h = n_rows.cpu().numpy().astype(np.int64)
        w = np.asarray(who, dtype=np.int64)
        nq, max_b = len(w), max(1, int(self.n_boxes.max()))
        m = np.fromiter((len(samples[s]) for s in who), dtype=np.int64, count=nq)
        off = np.zeros(B + 1, dtype=np.int64)
        sizes = np.zeros(B, dtype=np.int64)
        sizes[w] = m
        np.cumsum(sizes, out=off[1:])
        total_rows = int(off[-1])
        # the slot's upload: layout (every part starts on a 64-byte boundary), host views, device views
        desc_size = C.sizeof(_lib.PlaceQuery)
        at, lay = 0, {}
        for name, nbytes in (("desc", nq * desc_size), ("boxes", B * max_b * 80), ("off", (B + 1) * 8), ("who", nq * 8), ("need", B * 4),
                             ("open", B * 4), ("rows", total_rows * 40)):
            lay[name] = (at, nbytes)
            at += (nbytes + 63) & ~63
        up_pin, host, dev = self._room(at)
        hv = lambda name, dtype: host[lay[name][0]:lay[name][0] + lay[name][1]].view(dtype)
        dv = lambda name, dtype: dev[lay[name][0]:lay[name][0] + lay[name][1]].view(dtype)
        dptr = lambda name: np.uint64(dev.data_ptr() + lay[name][0])
        hv("boxes", np.float64).reshape(B, max_b, 10)[:] = self.boxes_h[:, :max_b]
        hv("off", np.int64)[:] = off
        hv("who", np.int64)[:] = w
        hv("need", np.int32)[:] = np.asarray(min_points, dtype=np.int32)
        still = hv("open", np.int32)
        still[:] = 0
        still[w] = 1
        np.concatenate([samples[s] for s in who], axis=0, out=hv("rows", np.float64).reshape(total_rows, 5))
        d = hv("desc", np.uint8)
        d[:] = 0
        d = d.view(np.dtype(_lib.PlaceQuery))
        self._fill_descriptors(d, who, w, m, off, dptr("boxes"), max_b, dptr("rows"), all_ranges, rows, alive, annos, ok_labels, ok_maps,
                               flavours, chunk)
        dev[:at].copy_(up_pin[:at], non_blocking=True)
        pb = PlaceBatch({"desc": d, "d_desc": dv("desc", torch.uint8), "m": m, "max_boxes": max(1, int(d["n_boxes"].max())),
                         "max_n_scene": int(d["n_scene"].max()), "max_n_orig": int(d["n_orig"].max()),
                         "keep": (rows, all_ranges, alive, dev)}, cand_cap=chunk, device=batch.device, packed=True)
        return self._try_candidates(pb, who, chunk, annos, rotation, n_poss, dv("off", torch.int64), dv("need", torch.int32),
                                    dv("who", torch.int64), dv("open", torch.int32))

    def _fill_descriptors(self, d, who, w, m, off, boxes_ptr, max_b, rows_ptr, all_ranges, rows, alive, annos, ok_labels, ok_maps,
                          flavours, chunk):
        """The descriptors of a slot's queries, all fields of all queries at once, written where they are uploaded from (the
        per-query form, ``_fill_query``, spent 8 of an insert slot's 11 ms on 256 frames in ctypes field stores)."""
        batch = self.batch
        cap = batch.cap
        u = w.astype(np.uint64)
        if alive is not None:                                            # the scene where it stands in the batch (R3D_PQ_SCENE_SLAB)
            d["scene"] = np.uint64(batch.xyzi.data_ptr()) + u * np.uint64(cap * 16)
            d["scene_label"] = np.uint64(batch.label.data_ptr()) + u * np.uint64(cap * 4)
            d["scene_alive"] = np.uint64(alive.data_ptr()) + u * np.uint64((cap // 64) * 8)
            d["scene_tail_ref"] = np.uint64(batch.tail_ref.data_ptr()) + u * np.uint64(batch.log_cap * 4)
            d["scene_log5"] = np.uint64(batch.log5.data_ptr()) + u * np.uint64(batch.log_cap * 40)
            d["scene_head"] = self.n_head_arr[w]
            d["flavour"] = _lib.PQ_SCENE_SLAB
        else:
            d["scene"] = np.uint64(rows.data_ptr()) + u * np.uint64(cap * 32)
        if self.slab:                                                    # the original cloud: the same slab's first rows (R3D_PQ_ORIG_SLAB)
            d["orig"] = np.uint64(batch.xyzi.data_ptr()) + u * np.uint64(cap * 16)
            d["orig_label"] = np.uint64(batch.label.data_ptr()) + u * np.uint64(cap * 4)
            d["flavour"] |= _lib.PQ_ORIG_SLAB
        else:
            d["orig"] = np.uint64(self.orig_rows.data_ptr()) + u * np.uint64(cap * 32)
        d["boxes"] = boxes_ptr + u * np.uint64(max_b * 80)
        d["sample"] = rows_ptr + off[w].astype(np.uint64) * np.uint64(40)
        d["map"] = self.map_ptr[w]
        d["scene_ranges"] = np.uint64(all_ranges.data_ptr()) + u * np.uint64((cap // 64) * 8)
        d["orig_ranges"] = np.uint64(self.orig_ranges_all.data_ptr()) + u * np.uint64((cap // 64) * 8)
        d["n_scene"], d["n_orig"] = self.n_scene_h[w], self.n_orig_arr[w]
        d["scene_ld"] = d["orig_ld"] = 4
        d["scene_label_col"] = d["orig_label_col"] = 3
        d["n_boxes"] = self.n_boxes[w]
        d["m"] = m
        d["map_rows"], d["map_cols"] = self.map_shape[w, 0], self.map_shape[w, 1]
        # placement labels / map codes: a handful of distinct (class) combinations per batch -- each is encoded once and
        # written to all its queries at a time (the per-query loop cost 0.4 ms of a slot on 256 frames)
        groups = {}
        for qi, s in enumerate(who):
            groups.setdefault((tuple(int(v) for v in ok_labels[s]), tuple(int(v) for v in ok_maps[s])), []).append(qi)
        for (ol, om), qis in groups.items():
            if len(ol) > _lib.PLACE_MAX_OK_LABELS:
                raise ValueError(f"at most {_lib.PLACE_MAX_OK_LABELS} placement labels per class")
            bits = [0, 0, 0, 0]
            for v in om:
                if 0 <= v <= 255:
                    bits[v >> 6] |= 1 << (v & 63)
            qis = np.asarray(qis, dtype=np.int64)
            d["n_ok_labels"][qis] = len(ol)
            lab = np.zeros(_lib.PLACE_MAX_OK_LABELS, dtype=np.int32)
            lab[:len(ol)] = ol
            d["ok_labels"][qis] = lab
            d["ok_map"][qis] = np.array(bits, dtype=np.uint64)
        if flavours:
            for qi, s in enumerate(who):
                if flavours[s]:
                    d["flavour"][qi] |= flavours[s].get("flavour", 0)
                    d["collide_label"][qi] = flavours[s].get("collide_label", 0)
                    d["collide_dz"][qi] = flavours[s].get("collide_dz", 0.0)
        d["anno"] = np.asarray([annos[s] for s in who], dtype=np.float64)[:, :10]
        d["pose"], d["map_move"] = self.pose_arr[w], self.move_arr[w]
        PlaceBatch.candidate_layout(d, m, chunk)

    def _try_candidates(self, pb, who, chunk, annos, rotation, n_poss, sample_off, need, who_t, still_open):
        """sample_off [B + 1] int64, need [B] int32, who_t [queries] int64, still_open [B] int32 (1 for the scenes of `who`):
        device tensors."""
        torch, batch = self.torch, self.batch
        B, nq = batch.B, len(who)
        accepted_at = torch.full((B,), -1, dtype=torch.int32, device=batch.device)
        first, new_slot = 0, True
        n_possible = torch.zeros(B, dtype=torch.int32, device=batch.device)
        while True:
            pb.run(first_cand=first)
            n_possible[who_t] = pb.n_possible
            # the window's candidates of every scene, in rotation order until one is accepted: ONE call (round 6;
            # r3d_batch_insert_first -- one r3d_batch_insert launch per candidate and the masks between them before)
            batch.insert_first_device(pb.cand, pb.total, sample_off, need, still_open, n_possible, first, chunk, accepted_at,
                                      replay_last=self.reference_rejected_state, new_slot=new_slot)
            new_slot = False
            still_open = still_open * (accepted_at < 0).to(torch.int32)
            more = bool(((n_possible > first + chunk).to(torch.int32) * still_open).any().item())   # one sync per window
            if not more:
                break
            first += chunk
        # per query: accepted candidate (-1: none), possible placements, the accepted one's rotation number and annotation, the
        # search's status; per scene: the batch's status and point count -- gathered here, two copies into pinned memory
        acc_w = accepted_at[who_t]
        j_t = acc_w.clamp(min=0).to(torch.int64)
        qi_t = self._arange[:nq]
        res = torch.empty((nq, 11), dtype=torch.float64, device=batch.device)
        res[:, 0], res[:, 1], res[:, 2], res[:, 3] = acc_w, n_possible[who_t], pb.rot_out[qi_t, j_t], pb.status
        res[:, 4:] = pb.anno_out[qi_t, j_t]
        self._down_f[:nq * 11].copy_(res.view(-1), non_blocking=True)
        self._down_i.copy_(torch.cat([batch.status, batch.n_total]), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        res_h = self._down_f[:nq * 11].numpy().reshape(nq, 11)
        st_b, self.n_total_h = self._down_i[:B].numpy(), self._down_i[B:].numpy().astype(np.int64)
        if self.slab:
            self.n_scene_h = self.n_total_h                              # (points of the slab, dead ones included)
        if batch._looked:
            batch.raise_on_status()                                      # (the first batch of a stream: its point order is noted)
        for s in np.nonzero(st_b)[0]:
            _lib.raise_status(int(st_b[s]), f"scene {s}")
        st = res_h[:, 3]
        if st.any():
            raise ValueError(f"placement search status {int(st[st != 0][0])} (see R3D_PS_*)")
        w = np.asarray(who, dtype=np.int64)
        got = res_h[:, 0] >= 0
        rot_w = np.where(got, res_h[:, 2], -1).astype(np.int64)
        for s, r, n in zip(who, rot_w.tolist(), res_h[:, 1].astype(np.int64).tolist()):
            rotation[s], n_poss[s] = r, n
        if got.any():
            # the accepted objects' boxes join their scenes' annotations (insertion.py:535): centre + quaternion of the
            # accepted placement, the sample's own extent
            ws, at = w[got], self.n_boxes[w[got]]
            if int(at.max()) >= self.boxes_h.shape[1]:
                self.boxes_h = np.concatenate([self.boxes_h, np.zeros((self.boxes_h.shape[0], 8, 10))], axis=1)
            ext = np.asarray([annos[s] for s in ws.tolist()], dtype=np.float64)[:, 7:10]
            self.boxes_h[ws, at, :7] = res_h[got, 4:]
            self.boxes_h[ws, at, 7:] = ext
            self.n_boxes[ws] += 1
        return rotation, n_poss
