"""Host side of the placement search (include/real3daug_hip.h, Level 3).

``find_places`` runs many (scene, sample) queries of the reference's ``find_possible_places``
(SS tools/find_spot.py:192-273) in one call of ``r3d_find_possible_places``.  The host only
packs descriptors; every step of the search runs in the HIP kernels and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def search_radii_sq():
    """radius**2 of the steps of correct_height's growing search that can still succeed
    (find_spot.py:119-140): the loop fails as soon as the *next* radius exceeds 5."""
    out, radius = [], 0.1
    while True:
        sq = radius ** 2
        radius += 0.1
        if radius > 5:
            return out
        out.append(sq)


def upload_map(rich_map, device):
    """The rich map as a uint8 tensor on the device (device tensors pass through)."""
    torch = _lib.require_gpu()
    if isinstance(rich_map, torch.Tensor):
        assert rich_map.dtype == torch.uint8 and rich_map.dim() == 2
        return rich_map.to(device).contiguous()
    m = np.asarray(rich_map)
    if m.dtype != np.uint8:
        if not np.array_equal(m, np.floor(m)) or m.min() < 0 or m.max() > 255:
            raise ValueError("the rich map must hold integer surface codes 0..255")
        m = m.astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(m)).to(device)


def chunk_ranges(rows):
    with _lib.on(rows):
        return _chunk_ranges(rows)


def _chunk_ranges(rows):
    """r3d_places_chunk_ranges of a device tensor of rows (x y first): float32 [chunks, 2].  float32 rows [n, 4] (a batch's
    slab): r3d_places_chunk_ranges_f32 -- the same ranges as for the float64 rows of the same points."""
    torch = _lib.require_gpu()
    n = rows.shape[0]
    out = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device=rows.device)
    if n and rows.dtype == torch.float32:
        assert rows.shape[1] == 4 and rows.is_contiguous()
        _lib.check(_lib.load().r3d_places_chunk_ranges_f32(rows.data_ptr(), n, out.data_ptr(), _lib.stream_ptr()),
                   "r3d_places_chunk_ranges_f32")
    elif n:
        _lib.check(_lib.load().r3d_places_chunk_ranges(rows.data_ptr(), n, rows.shape[1], out.data_ptr(),
                                                      _lib.stream_ptr()), "r3d_places_chunk_ranges")
    return out


class PlaceScene:
    """Device-resident inputs of one scene, shared by the queries that use it."""

    def __init__(self, point_cloud, original_pcl, scene_boxes, rich_map, map_move, transformation_matrix,
                 scene_label_col=None, orig_label_col=None, device="cuda:0", scene_ranges=None, orig_ranges=None):
        torch = _lib.require_gpu()
        self.device = device

        def dev64(a):
            if isinstance(a, torch.Tensor):
                return a.to(device=device, dtype=torch.float64).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)

        scene, orig = dev64(point_cloud), dev64(original_pcl)
        assert scene.dim() == 2 and orig.dim() == 2
        # scene_pcl is N x 9 with the label in column 7 (insertion.py:433), original_pcl N x 5 with it in
        # column 4; N x 4 rows are taken as already packed [x y z label] (SceneBatch.export_rows)
        def label_col(t, given):
            return given if given is not None else {9: 7, 5: 4, 4: 3}[t.shape[1]]
        scene_label_col, orig_label_col = label_col(scene, scene_label_col), label_col(orig, orig_label_col)
        # the search reads x y z and the label only: keep them as packed 32-byte rows
        self.scene = scene if scene.shape[1] == 4 else scene[:, [0, 1, 2, scene_label_col]].contiguous()
        self.orig = orig if orig.shape[1] == 4 else orig[:, [0, 1, 2, orig_label_col]].contiguous()
        self.scene_label_col = self.orig_label_col = 3
        self.scene_ranges = chunk_ranges(self.scene) if scene_ranges is None else scene_ranges
        self.orig_ranges = chunk_ranges(self.orig) if orig_ranges is None else orig_ranges
        boxes = np.ascontiguousarray(scene_boxes, dtype=np.float64).reshape(-1, 10)
        self.n_boxes = len(boxes)
        self.boxes = torch.from_numpy(boxes if len(boxes) else np.zeros((1, 10))).to(device)
        self.map = upload_map(rich_map, device)
        self.map_move = (float(np.asarray(map_move).reshape(-1)[0]), float(np.asarray(map_move).reshape(-1)[1]))
        self.pose = np.asarray(transformation_matrix, dtype=np.float64)[:2, :4].reshape(8).copy()


def scene_view(scene_rows, orig_rows, boxes_t, n_boxes, map_t, map_move, pose8, scene_ranges, orig_ranges):
    """A PlaceScene over device tensors that are already in the layout the search reads (packed
    [x y z label] rows, uploaded boxes / map, chunk ranges): no copies, no launches."""
    ps = PlaceScene.__new__(PlaceScene)
    ps.device = scene_rows.device
    ps.scene, ps.orig, ps.scene_label_col, ps.orig_label_col = scene_rows, orig_rows, 3, 3
    ps.boxes, ps.n_boxes, ps.map = boxes_t, int(n_boxes), map_t
    ps.map_move, ps.pose = map_move, pose8
    ps.scene_ranges, ps.orig_ranges = scene_ranges, orig_ranges
    return ps


def _fill_query(qd, scene, sample_t, anno10, ok_labels, ok_map_values, cand_cap, cand_off, cand_stride,
                flavour=0, collide_label=0, collide_dz=0.0):
    qd.scene, qd.orig = scene.scene.data_ptr(), scene.orig.data_ptr()
    qd.boxes, qd.sample, qd.map = scene.boxes.data_ptr(), sample_t.data_ptr(), scene.map.data_ptr()
    qd.scene_ranges = scene.scene_ranges.data_ptr() if scene.scene_ranges.numel() else None
    qd.orig_ranges = scene.orig_ranges.data_ptr() if scene.orig_ranges.numel() else None
    qd.n_scene, qd.n_orig = scene.scene.shape[0], scene.orig.shape[0]
    qd.scene_ld, qd.scene_label_col = scene.scene.shape[1], scene.scene_label_col
    qd.orig_ld, qd.orig_label_col = scene.orig.shape[1], scene.orig_label_col
    qd.n_boxes, qd.m = scene.n_boxes, sample_t.shape[0]
    qd.map_rows, qd.map_cols = scene.map.shape
    if len(ok_labels) > _lib.PLACE_MAX_OK_LABELS:
        raise ValueError("at most 8 placement labels per class")
    qd.n_ok_labels = len(ok_labels)
    for i, v in enumerate(ok_labels):
        qd.ok_labels[i] = int(v)
    bits = [0, 0, 0, 0]
    for v in ok_map_values:
        if 0 <= int(v) <= 255:
            bits[int(v) >> 6] |= 1 << (int(v) & 63)
    for i in range(4):
        qd.ok_map[i] = bits[i]
    for i in range(10):
        qd.anno[i] = float(anno10[i])
    for i in range(8):
        qd.pose[i] = float(scene.pose[i])
    qd.map_move[0], qd.map_move[1] = scene.map_move
    qd.cand_cap, qd.cand_off, qd.cand_stride = int(cand_cap), int(cand_off), int(cand_stride)
    qd.flavour, qd.collide_label, qd.collide_dz = int(flavour), int(collide_label), float(collide_dz)


class PlaceBatch:
    """Descriptors, outputs and workspace of a set of queries on the device; ``run`` launches the
    search (asynchronously, on the current stream), ``results`` downloads and unpacks."""

    def __init__(self, queries, cand_cap=360, device="cuda:0", packed=False):
        """packed=False: the candidate clouds of a query are contiguous ([cand_cap][m][5] per query).
        packed=True: candidate j of all queries forms one packed sample list (rows of query q at
        ``sample_off[q]``) at ``cand[j * total:]`` -- the layout ``r3d_batch_insert`` takes."""
        torch = _lib.require_gpu()
        self.lib = _lib.load()
        self.device, self.cand_cap = device, int(cand_cap)
        if isinstance(queries, dict):                       # descriptors already packed (PlacedInserter): no per-query work
            return self._from_arrays(queries, packed)
        self.nq = nq = len(queries)
        if nq == 0:
            raise ValueError("no queries")
        self.queries = queries
        self.samples = [q["sample"] if isinstance(q["sample"], torch.Tensor) else
                        torch.from_numpy(np.ascontiguousarray(q["sample"], dtype=np.float64).reshape(-1, 5)).to(device)
                        for q in queries]
        descs = (_lib.PlaceQuery * nq)()
        cand_off, self.offs, self.packed = 0, [], packed
        self.total = sum(s.shape[0] for s in self.samples) * 5
        for i, q in enumerate(queries):
            m = self.samples[i].shape[0]
            if m == 0:
                raise ValueError("empty sample")
            _fill_query(descs[i], q["scene"], self.samples[i], q["anno"], q["ok_labels"], q["ok_map"], cand_cap, cand_off,
                        self.total if packed else m * 5, q.get("flavour", 0), q.get("collide_label", 0),
                        q.get("collide_dz", 0.0))
            self.offs.append(cand_off)
            cand_off += m * 5 if packed else self.cand_cap * m * 5
        if packed:
            cand_off = self.total * self.cand_cap
        self.d_desc = torch.from_numpy(np.frombuffer(descs, dtype=np.uint8).copy()).to(device)
        self._outputs(cand_off, max(q["scene"].n_boxes for q in queries), max(q["scene"].scene.shape[0] for q in queries),
                      max(q["scene"].orig.shape[0] for q in queries), max(s.shape[0] for s in self.samples))

    def _from_arrays(self, a, packed):
        """a: {"desc": structured array of PlaceQuery records (cand_off / cand_stride / cand_cap filled here), "m": sample sizes,
        "max_boxes", "max_n_scene", "max_n_orig", "keep": tensors the descriptors point into}; packed layout only."""
        torch = _lib.require_gpu()
        assert packed
        desc, m = a["desc"], np.asarray(a["m"], dtype=np.int64)
        self.nq = len(desc)
        if self.nq == 0:
            raise ValueError("no queries")
        if (m <= 0).any():
            raise ValueError("empty sample")
        self.queries, self.samples, self.packed = None, None, True
        self.sample_sizes = m
        self.total = int(m.sum()) * 5
        offs = np.zeros(self.nq, dtype=np.int64)
        offs[1:] = np.cumsum(m[:-1] * 5)
        self.offs = [int(x) for x in offs]
        self._keep = a.get("keep")
        if a.get("d_desc") is not None:                     # uploaded by the caller, candidate_layout() applied
            self.d_desc = a["d_desc"]
        else:
            self.candidate_layout(desc, m, self.cand_cap)
            self.d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(self.device)
        self._outputs(self.total * self.cand_cap, int(a["max_boxes"]), int(a["max_n_scene"]), int(a["max_n_orig"]), int(m.max()))

    @staticmethod
    def candidate_layout(desc, m, cand_cap):
        """cand_cap / cand_off / cand_stride of packed descriptors (candidate j of all queries = one packed sample list)."""
        m = np.asarray(m, dtype=np.int64)
        offs = np.zeros(len(m), dtype=np.int64)
        offs[1:] = np.cumsum(m[:-1] * 5)
        desc["cand_cap"], desc["cand_off"], desc["cand_stride"] = int(cand_cap), offs, int(m.sum()) * 5

    def _outputs(self, cand_off, max_boxes, max_n_scene, max_n_orig, max_m):
        torch = _lib.require_gpu()
        device, nq = self.device, self.nq
        rot = _lib.PLACE_ROTATIONS
        self.flags = torch.zeros((nq, rot), dtype=torch.uint8, device=device)
        self.n_possible = torch.zeros(nq, dtype=torch.int32, device=device)
        self.rot_out = torch.zeros((nq, rot), dtype=torch.int32, device=device)
        self.anno_out = torch.zeros((nq, rot, 7), dtype=torch.float64, device=device)
        self.cand = torch.empty(max(cand_off, 1), dtype=torch.float64, device=device)
        self.status = torch.zeros(nq, dtype=torch.int32, device=device)
        self.max_boxes = max_boxes
        self.ws_bytes = self.lib.r3d_places_workspace_bytes(nq, self.max_boxes)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        radii = search_radii_sq()
        self.n_radii = len(radii)
        self.radii_c = (C.c_double * len(radii))(*radii)
        self.max_n_scene, self.max_n_orig, self.max_m = max_n_scene, max_n_orig, max_m
        self.first_cand = 0

    @_lib.on_own_device
    def run(self, first_cand=0):
        self.first_cand = int(first_cand)
        _lib.check(self.lib.r3d_find_possible_places(
            self.d_desc.data_ptr(), self.nq, self.max_n_scene, self.max_n_orig, self.max_m, self.max_boxes,
            self.radii_c, self.n_radii, self.flags.data_ptr(), self.n_possible.data_ptr(), self.rot_out.data_ptr(),
            self.anno_out.data_ptr(), self.cand.data_ptr(), self.first_cand, self.status.data_ptr(),
            self.ws.data_ptr(), self.ws_bytes, _lib.stream_ptr()), "r3d_find_possible_places")
        return self

    @_lib.on_own_device
    def results(self):
        import torch
        torch.cuda.synchronize()
        flags_h, n_h, rot_h = self.flags.cpu().numpy(), self.n_possible.cpu().numpy(), self.rot_out.cpu().numpy()
        anno_h, status_h, cand_h = self.anno_out.cpu().numpy(), self.status.cpu().numpy(), self.cand.cpu().numpy()
        out = []
        for i in range(self.nq):
            if status_h[i] & _lib.PS_BAD_DESCRIPTOR:
                raise ValueError(f"query {i}: the descriptor cannot be followed (R3D_PS_BAD_DESCRIPTOR, include/real3daug_hip.h)")
            if status_h[i] & _lib.PS_NONFINITE:
                raise ValueError(f"query {i}: NaN/Inf in the sample, its box or the pose")
            if status_h[i] & _lib.PS_SURFACE_OVERFLOW:
                raise ValueError(f"query {i}: more than {_lib.PLACE_SURFACE_CAP} surface points inside the search radius")
            n, m = int(n_h[i]), self.samples[i].shape[0]
            k = max(0, min(n - self.first_cand, self.cand_cap))
            if self.packed:
                clouds = np.stack([cand_h[j * self.total + self.offs[i]: j * self.total + self.offs[i] + m * 5].reshape(m, 5)
                                   for j in range(k)]) if k else np.zeros((0, m, 5))
            else:
                clouds = cand_h[self.offs[i]: self.offs[i] + k * m * 5].reshape(k, m, 5).copy()
            out.append({"flags": flags_h[i], "rotations": rot_h[i, :n].copy(), "anno": anno_h[i, :n].copy(),
                        "clouds": clouds, "status": int(status_h[i])})
        return out


def find_places(queries, cand_cap=360, first_cand=0, device="cuda:0"):
    """queries: list of dicts with keys ``scene`` (PlaceScene), ``sample`` (M x 5 float64),
    ``anno`` (10 floats: centre, quaternion xyzw, length, width, height), ``ok_labels``,
    ``ok_map`` (allowed map values); for the object-detection flavour also ``flavour``
    (``_lib.PQ_*`` bits), ``collide_label``, ``collide_dz``.  Returns per query a dict: ``flags`` uint8[360],
    ``rotations`` int32[n], ``anno`` float64[n,7], ``clouds`` float64[k,M,5] (placements
    ``first_cand`` .. ``first_cand + cand_cap``), ``status``."""
    if len(queries) == 0:
        return []
    return PlaceBatch(queries, cand_cap, device).run(first_cand).results()
