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

import numpy as np

from . import _lib
from .places import PlaceBatch, chunk_ranges, scene_view, upload_map


class PlacedInserter:
    def __init__(self, batch, rich_maps, map_moves, poses, scene_boxes, reference_rejected_state=False):
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
            slab = torch.from_numpy(np.ascontiguousarray(np.stack(rich_maps))).to(batch.device, non_blocking=True)
            self.maps = [slab[s] for s in range(B)]
        else:
            self.maps = [upload_map(m, batch.device) for m in rich_maps]
        self.moves = [(float(np.asarray(m).reshape(-1)[0]), float(np.asarray(m).reshape(-1)[1])) for m in map_moves]
        self.poses = [np.asarray(p, dtype=np.float64)[:2, :4].reshape(8).copy() for p in poses]
        self.boxes = [np.asarray(b, dtype=np.float64).reshape(-1, 10) for b in scene_boxes]
        # original_pcl (insertion.py:360): the clouds as loaded, as packed float64 rows
        # (the counts `load` wrote into the pinned staging, when the batch was loaded that way: no wait for the device)
        pin = getattr(batch, "_pin", None)
        n0 = pin["n"].numpy() if pin is not None and getattr(batch, "_loaded_from_staging", False) else batch.n_points.cpu().numpy()
        self.n_orig = [int(v) for v in n0]
        self.orig_rows = torch.cat([batch.xyzi[:, :, :3].to(torch.float64),
                                    (batch.label.to(torch.int64) & 0xFFFF).to(torch.float64)[:, :, None]], dim=2).contiguous()
        # 64-point chunks of a scene are whole chunks of the slab when its stride is a multiple of 64
        self.chunked = batch.cap % 64 == 0
        if self.chunked:
            r = chunk_ranges(self.orig_rows.view(B * batch.cap, 4)).view(B, batch.cap // 64, 2)
            # (a scene's last chunk may take in rows past its end: its range only gets wider, which is safe)
            self.orig_ranges_all = r
            self.orig_ranges = [r[s, :(self.n_orig[s] + 63) // 64] for s in range(B)]
        else:
            self.orig_ranges = [chunk_ranges(self.orig_rows[s, :self.n_orig[s]]) for s in range(B)]
        # what the descriptors of a slot are packed from, per scene, as arrays (insert_slot fills all queries at once)
        self.map_ptr = np.array([m.data_ptr() for m in self.maps], dtype=np.uint64)
        self.map_shape = np.array([m.shape for m in self.maps], dtype=np.int32)
        self.move_arr = np.array(self.moves, dtype=np.float64)
        self.pose_arr = np.array(self.poses, dtype=np.float64)
        self.n_orig_arr = np.array(self.n_orig, dtype=np.int64)

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

    def _insert_slot(self, samples, annos, ok_labels, ok_maps, min_points, chunk, flavours):
        torch, batch = self.torch, self.batch
        B = batch.B
        rows, n_rows = batch.export_rows()
        n_rows_h = n_rows.cpu().numpy()
        who = [s for s in range(B) if samples[s] is not None and len(samples[s])]
        rotation, n_poss = [-1] * B, [0] * B
        if not who:
            return rotation, n_poss
        # one upload for the boxes of all scenes, one launch for the chunk ranges of all current clouds
        max_b = max(1, max(len(b) for b in self.boxes))
        boxes_h = np.zeros((B, max_b, 10))
        for s in range(B):
            boxes_h[s, :len(self.boxes[s])] = self.boxes[s]
        boxes_d = torch.from_numpy(boxes_h).to(batch.device)
        if self.chunked:
            all_ranges = chunk_ranges(rows.view(B * batch.cap, 4)).view(B, batch.cap // 64, 2)
        in_who = set(who)
        smp_rows, smp_off = batch.pack_samples([samples[s] if s in in_who else None for s in range(B)])
        smp_off_h = smp_off.cpu().numpy()
        if self.chunked:
            pb = self._pack_slot(who, rows, n_rows_h, boxes_d, max_b, all_ranges, smp_rows, smp_off_h, annos, ok_labels, ok_maps,
                                 flavours, chunk)
            return self._try_candidates(pb, who, min_points, chunk, annos, rotation, n_poss)
        queries = []
        for s in who:
            n = int(n_rows_h[s])
            if self.chunked:
                rng_s = all_ranges[s, :(n + 63) // 64]
            else:
                rng_s = chunk_ranges(rows[s, :n])
            scene = scene_view(rows[s, :n], self.orig_rows[s, :self.n_orig[s]], boxes_d[s], len(self.boxes[s]), self.maps[s],
                               self.moves[s], self.poses[s], rng_s, self.orig_ranges[s])
            queries.append({"scene": scene, "sample": smp_rows[int(smp_off_h[s]):int(smp_off_h[s + 1])], "anno": annos[s],
                            "ok_labels": ok_labels[s], "ok_map": ok_maps[s], **((flavours[s] or {}) if flavours else {})})
        pb = PlaceBatch(queries, cand_cap=chunk, device=batch.device, packed=True)
        pb.sample_sizes = np.array([q.shape[0] for q in pb.samples], dtype=np.int64)
        return self._try_candidates(pb, who, min_points, chunk, annos, rotation, n_poss)

    def _pack_slot(self, who, rows, n_rows_h, boxes_d, max_b, all_ranges, smp_rows, smp_off_h, annos, ok_labels, ok_maps, flavours,
                   chunk):
        """The descriptors of a slot's queries, all fields of all queries at once (the per-query form, ``_fill_query``,
        spent 8 of an insert slot's 11 ms on 256 frames in ctypes field stores)."""
        batch = self.batch
        w = np.asarray(who, dtype=np.int64)
        nq, cap = len(w), batch.cap
        d = np.zeros(nq, dtype=np.dtype(_lib.PlaceQuery))
        u = w.astype(np.uint64)
        d["scene"] = np.uint64(rows.data_ptr()) + u * np.uint64(cap * 32)
        d["orig"] = np.uint64(self.orig_rows.data_ptr()) + u * np.uint64(cap * 32)
        d["boxes"] = np.uint64(boxes_d.data_ptr()) + u * np.uint64(max_b * 80)
        d["sample"] = np.uint64(smp_rows.data_ptr()) + smp_off_h[w].astype(np.uint64) * np.uint64(40)
        d["map"] = self.map_ptr[w]
        d["scene_ranges"] = np.uint64(all_ranges.data_ptr()) + u * np.uint64((cap // 64) * 8)
        d["orig_ranges"] = np.uint64(self.orig_ranges_all.data_ptr()) + u * np.uint64((cap // 64) * 8)
        d["n_scene"], d["n_orig"] = n_rows_h[w], self.n_orig_arr[w]
        d["scene_ld"] = d["orig_ld"] = 4
        d["scene_label_col"] = d["orig_label_col"] = 3
        d["n_boxes"] = [len(self.boxes[s]) for s in who]
        m = (smp_off_h[w + 1] - smp_off_h[w]).astype(np.int64)
        d["m"] = m
        d["map_rows"], d["map_cols"] = self.map_shape[w, 0], self.map_shape[w, 1]
        for qi, s in enumerate(who):
            ol = ok_labels[s]
            if len(ol) > _lib.PLACE_MAX_OK_LABELS:
                raise ValueError("at most 8 placement labels per class")
            d["n_ok_labels"][qi] = len(ol)
            d["ok_labels"][qi, :len(ol)] = ol
            bits = [0, 0, 0, 0]
            for v in ok_maps[s]:
                if 0 <= int(v) <= 255:
                    bits[int(v) >> 6] |= 1 << (int(v) & 63)
            d["ok_map"][qi] = bits
            if flavours and flavours[s]:
                d["flavour"][qi] = flavours[s].get("flavour", 0)
                d["collide_label"][qi] = flavours[s].get("collide_label", 0)
                d["collide_dz"][qi] = flavours[s].get("collide_dz", 0.0)
        d["anno"] = np.array([np.asarray(annos[s], dtype=np.float64)[:10] for s in who])
        d["pose"], d["map_move"] = self.pose_arr[w], self.move_arr[w]
        return PlaceBatch({"desc": d, "m": m, "max_boxes": max(1, int(d["n_boxes"].max())), "max_n_scene": int(d["n_scene"].max()),
                           "max_n_orig": int(d["n_orig"].max()), "keep": (rows, boxes_d, all_ranges, smp_rows)},
                          cand_cap=chunk, device=batch.device, packed=True)

    def _try_candidates(self, pb, who, min_points, chunk, annos, rotation, n_poss):
        torch, batch = self.torch, self.batch
        B = batch.B
        sizes = np.zeros(B, dtype=np.int64)
        sizes[who] = pb.sample_sizes
        off = np.zeros(B + 1, dtype=np.int64)
        off[1:] = np.cumsum(sizes)
        sample_off = torch.from_numpy(off).to(batch.device)
        need = torch.from_numpy(np.asarray(min_points, dtype=np.int32)).to(batch.device)
        who_t = torch.tensor(who, dtype=torch.int64, device=batch.device)
        still_open = torch.zeros(B, dtype=torch.int32, device=batch.device)
        still_open[who_t] = 1
        accepted_at = torch.full((B,), -1, dtype=torch.int32, device=batch.device)
        first, new_slot = 0, True
        while True:
            pb.run(first_cand=first)
            n_possible = torch.zeros(B, dtype=torch.int32, device=batch.device)
            n_possible[who_t] = pb.n_possible
            for j in range(chunk):
                active = still_open * (n_possible > first + j).to(torch.int32)
                nv, acc = batch.insert_device(pb.cand[j * pb.total:], sample_off, need, active, new_slot=new_slot)
                new_slot = False
                got = acc * active
                accepted_at = torch.where(got > 0, torch.full_like(accepted_at, first + j), accepted_at)
                still_open = still_open * (1 - got)
                if self.reference_rejected_state:
                    # the sample's LAST candidate, rejected with a visible part: the driver keeps the scene it has culled
                    # (the candidate is replayed into the batch's shadow; nothing of the scene changes)
                    # (also one rejected WITHOUT a visible point: a closing-filled hole of the sample in front of the scene is a
                    # visible pixel all the same and culls there, insertion.py:467-473)
                    last_rejected = active * still_open * (n_possible == first + j + 1).to(torch.int32)
                    batch.insert_device(pb.cand[j * pb.total:], sample_off, torch.full_like(need, -1), last_rejected, new_slot=False)
            more = bool(((n_possible > first + chunk).to(torch.int32) * still_open).any().item())   # one sync per chunk
            if not more:
                break
            first += chunk
        acc_h, n_h = accepted_at.cpu().numpy(), n_possible.cpu().numpy()
        rot_h, anno_h = pb.rot_out.cpu().numpy(), pb.anno_out.cpu().numpy()
        batch.raise_on_status()
        st = pb.status.cpu().numpy()
        if st.any():
            raise ValueError(f"placement search status {st[st != 0][0]} (see R3D_PS_*)")
        for qi, s in enumerate(who):
            n_poss[s] = int(n_h[s])
            j = int(acc_h[s])
            if j >= 0:
                rotation[s] = int(rot_h[qi, j])
                box = np.concatenate([anno_h[qi, j], np.asarray(annos[s], dtype=np.float64)[7:10]])
                self.boxes[s] = np.vstack([self.boxes[s], box[None, :]])       # insertion.py:535
        return rotation, n_poss
