"""File-to-file driver around the batched hot path.

Reads frames the way the reference's dataset classes do (``velodyne/*.bin`` float32 N x 4 +
``labels/*.label`` uint32, SS tools/datasets.py:51-56; OD tools/datasets.py:56-62), runs
``SceneBatch`` on batches of them, and writes ``velodyne/{f}.bin``, ``labels/{f}.label`` (not for
KITTI object detection) and ``check/{f}.bin`` byte for byte like ``save_data``
(SS tools/datasets.py:72-91, OD :76-95).  What the reference's driver decides on the host stays
with the caller: ``candidates_for(i)`` returns, for frame i, the ordered placement candidates of
every insert (``find_possible_places`` output, find_spot.py:192) and the ``min_points`` of their
classes.

Reading, computing and writing overlap: a reader thread parses the files of batch i+1 and a
writer thread stores the results of batch i-1 while batch i is on the GPU; the device descriptor
(``SceneBatch``) is allocated once and re-used.  ``run`` / ``run_placed`` keep the candidate loop
(several placements per insert) and use one stream; ``run_streamed`` (one placement per insert) also
overlaps upload, kernels and download over several lanes with pinned buffers (``streaming.py``).
The device leg is the HIP path and nothing else: there is no fallback and no parameter that selects another one.
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
import time

import numpy as np

from .Real3DAug.tools.datasets import read_frame, read_frame_waymo, write_frame, write_frame_waymo


class Frame:
    """One input frame: where its scan and labels are, and the name the outputs get."""

    def __init__(self, velodyne_file, label_file, name=None):
        self.velodyne_file, self.label_file = velodyne_file, label_file
        self.name = name or os.path.splitext(os.path.basename(velodyne_file))[0]


def _outputs_exist(output_path, folder, name, write_labels, waymo=False):
    if waymo:
        return all(os.path.exists(os.path.join(output_path, folder, sub, f"{name}.npy")) for sub in ("lidar", "labels_v3_2", "check"))
    subs = ("velodyne", "check") + (("labels",) if write_labels else ())
    ext = {"velodyne": "bin", "check": "bin", "labels": "label"}
    return all(os.path.exists(os.path.join(output_path, folder, sub, f"{name}.{ext[sub]}")) for sub in subs)


class AugmentPipeline:
    def __init__(self, output_path, folder, dataset="semantic", batch_size=64, device="cuda:0",
                 collapse_labels_to_road=None, resume=True, reference_rejected_state=False):
        """dataset: "semantic" (SemanticKITTI: labels written, 5-column check file) or "kitti"
        (object detection: labels collapsed to {Road, 1} before use, OD insertion.py:353-355,
        no label file, 4-column check file).
        reference_rejected_state (``run_placed``): go on as the reference's driver does after a sample whose candidates were
        all rejected -- with the scene the last of them has culled (``PlacedInserter``; a slot dict may carry
        ``last_try: False`` when another sample for the same object follows)."""
        assert dataset in ("semantic", "kitti", "waymo")
        # "waymo": frames are lidar/{f}.npy (+ labels_v3_2/, SS tools/datasets.py:239-270), clouds are float64 after the
        # LiDAR offset is subtracted and go through r3d_batch_begin_f64; outputs are the three .npy files of :287-301
        self.output_path, self.folder, self.dataset = output_path, folder, dataset
        self.batch_size, self.device, self.resume = int(batch_size), device, resume
        self.check_cols = 5 if dataset == "semantic" else 4
        self.write_labels = dataset == "semantic"
        self.road_label = 40 if collapse_labels_to_road is None and dataset == "kitti" else collapse_labels_to_road
        self.waymo = dataset == "waymo"
        self.process = self._process_waymo if self.waymo else self._process_hip      # the device leg of `run`
        self.reference_rejected_state = bool(reference_rejected_state)
        self._batches = {}
        self._lane_batches = threading.local()       # run(lanes > 1): every worker thread keeps its own device batches

    # -- the GPU leg ------------------------------------------------------------------------------
    def _cache(self):
        """The device batches this thread re-uses (one set per worker thread of ``run``)."""
        c = getattr(self._lane_batches, "cache", None)
        if c is None:
            c = self._lane_batches.cache = {} if threading.current_thread() is not threading.main_thread() else self._batches
        return c

    def _process_hip(self, scenes, candidates, min_points):
        from .batch import augment_batch
        return augment_batch(scenes, candidates, min_points, device=self.device, check_cols=self.check_cols,
                             reuse=self._cache())

    def _process_waymo(self, scenes5, candidates, min_points):
        """Float64 frames (N x 5 rows) through ``SceneBatch.begin_f64``; results = (merged N' x 5, added M x 5, None)."""
        from .batch import SceneBatch
        B = len(scenes5)
        grow = max(sum(max((len(x) for x in slot), default=0) for slot in c) for c in candidates)
        n_max = max(len(x) for x in scenes5)
        cache = self._cache()
        batch = cache.get(("waymo", B))
        if batch is None or batch.cap < n_max + grow or batch.log_cap < n_max + grow:
            batch = SceneBatch(B, int((n_max + grow) * 1.05) + 64, int((n_max + grow) * 1.05) + 64, device=self.device)
            cache[("waymo", B)] = batch
        batch.begin_f64(scenes5)
        accepted = batch.run_inserts(candidates, min_points)
        batch.raise_on_status()
        return [(m, a, None) for m, a in batch.results_f64()], accepted

    # -- reading / writing -------------------------------------------------------------------------
    def _read(self, frame):
        if self.waymo:
            return read_frame_waymo(frame.velodyne_file, frame.label_file)[0]
        xyzi, label, _ = read_frame(frame.velodyne_file, frame.label_file)
        if self.road_label is not None:                       # OD insertion.py:353-355
            label = np.where(label == self.road_label, self.road_label, 1).astype(np.uint32)
        return np.ascontiguousarray(xyzi), np.ascontiguousarray(label.astype(np.uint32))

    def _read_batch(self, batch_frames):
        """The frames of a batch: .bin / .label files by native threads (``r3d_host_read_frames``; NumPy's fromfile
        holds the interpreter lock, one frame at a time) into one slab, handed out as views."""
        if self.waymo or any(f.label_file is None for f in batch_frames):
            return [self._read(f) for f in batch_frames]
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        B = len(batch_frames)
        cap = max(max(os.path.getsize(f.velodyne_file) // 16 for f in batch_frames), 1)
        xyzi, label = np.empty((B, cap, 4), dtype=np.float32), np.empty((B, cap), dtype=np.uint32)
        n = np.zeros(B, dtype=np.int32)
        enc = lambda paths: (C.c_char_p * B)(*[str(p).encode() for p in paths])
        _lib.check(lib.r3d_host_read_frames(enc([f.velodyne_file for f in batch_frames]), enc([f.label_file for f in batch_frames]),
                                            B, cap, xyzi.ctypes.data, label.ctypes.data, n.ctypes.data,
                                            -1 if self.road_label is None else int(self.road_label), 16), "r3d_host_read_frames")
        return [(xyzi[s, :n[s]], label[s, :n[s]]) for s in range(B)]

    # -- placement search + merge (the whole per-frame body of insertion.py:352-549) -------------------
    def _process_placed(self, scenes, infos, slots):
        """scenes as in _process_hip; infos[s] = (rich_map, map_move, pose 4x4, boxes k x 10);
        slots[s] = list of dicts {sample, anno (10 floats), ok_labels, ok_map, min_points, name}."""
        from .batch import SceneBatch
        from .placed import PlacedInserter
        B = len(scenes)
        k_max = max(len(sl) for sl in slots)
        grow = sum(max((len(sl[k]["sample"]) if k < len(sl) else 0) for sl in slots) for k in range(k_max))
        cap = max(len(x) for x, _ in scenes) + grow
        cache = self._cache()
        batch = cache.get(("placed", B))
        if batch is None or batch.cap < cap or batch.log_cap < max(grow, 1):
            batch = SceneBatch(B, int(cap * 1.05) + 64, max(grow, 1) * 2, device=self.device)
            cache[("placed", B)] = batch
        batch.load(scenes)
        batch.begin()
        ins = PlacedInserter(batch, [i[0] for i in infos], [i[1] for i in infos], [i[2] for i in infos], [i[3] for i in infos],
                             reference_rejected_state=getattr(self, "reference_rejected_state", False))
        chosen = [[] for _ in range(B)]
        for k in range(k_max):
            have = [sl[k] if k < len(sl) else None for sl in slots]
            rot, _ = ins.insert_slot([h["sample"] if h else None for h in have], [h["anno"] if h else None for h in have],
                                     [h["ok_labels"] if h else None for h in have], [h["ok_map"] if h else None for h in have],
                                     [h["min_points"] if h else 0 for h in have],
                                     last_try=[bool(h.get("last_try", True)) if h else True for h in have])
            for s in range(B):
                if have[s]:
                    chosen[s].append(rot[s])
        # (the delta instead of the merged clouds: the frames are still in the pinned staging `load` filled)
        return batch.results(delta_check_cols=self.check_cols), chosen

    def run_placed(self, frames, scene_info_for, slots_for, lanes=1):
        """Frames with the placement search in the loop: scene_info_for(i) -> (rich_map, map_move,
        pose, boxes) of frame i, slots_for(i) -> the samples to insert (see _process_placed).  Writes
        what ``run`` writes plus ``added_objects/{f}.txt`` with one line per inserted object,
        '<name> with rotation: <r>' (insertion.py:513)."""
        infos = {}                          # filled by the reader thread, taken by the GPU leg of the frame's batch

        def candidates_and_id(i):
            infos[i] = scene_info_for(i)
            return slots_for(i), i          # (the frame's index travels where `run` carries min_points)

        def process_placed(scenes, cands, ids):
            results, chosen = self._process_placed(scenes, [infos.pop(i) for i in ids], cands)
            os.makedirs(os.path.join(self.output_path, self.folder, "added_objects"), exist_ok=True)
            for i, sl, ch in zip(ids, cands, chosen):
                with open(os.path.join(self.output_path, self.folder, "added_objects", f"{frames[i].name}.txt"), "w") as fh:
                    for h, r in zip(sl, ch):
                        if r > 0:
                            fh.write(f"{h.get('name', 'object')} with rotation: {r}\n")
            return results, chosen

        saved = self.process
        self.process = process_placed
        try:
            return self.run(frames, candidates_and_id, lanes=lanes)
        finally:
            self.process = saved

    def run_streamed(self, frames, inserts_for, lanes=None, label_2_for=None, pack_threads=None, io_threads=None, delta=True):
        """Like ``run`` for ONE placement per insert: inserts_for(i) -> (samples, min_points) with
        samples[k] = M x 5 float64 (or None).  Batches go through ``StreamedAugmenter`` lanes: pinned
        buffers, native packing, upload / kernels / download of consecutive batches overlapped; only the delta
        of a batch comes back from the device, the merged clouds are put together on the host.

        ``io_threads`` threads read the frames of the next batches ahead of the GPU (a reader thread keeps two
        batches in flight) and write the files of a finished batch straight from the lane's buffers -- the lane
        is handed back when its files are on disk, the other lanes keep the GPU busy meanwhile."""
        from concurrent.futures import ThreadPoolExecutor
        from .streaming import StreamedAugmenter
        # host threads: as many as this process may use, at most 16 (a rank of a multi-GPU node is bound to its slice of the
        # host's cores first, affinity.bind_rank: eight ranks x sixteen unbound threads is what made the host the bottleneck)
        fair = max(1, min(16, len(os.sched_getaffinity(0))))
        pack_threads = fair if pack_threads is None else max(1, int(pack_threads))
        io_threads = fair if io_threads is None else max(1, int(io_threads))
        if lanes is None:
            # (the environment only stands in for an argument that was not given; read once, before anything is started)
            try:
                lanes = int(os.environ.get("R3D_STREAM_LANES", "3"))
            except ValueError:
                raise ValueError("R3D_STREAM_LANES must be an integer") from None
        lanes = max(1, int(lanes))
        todo = [i for i, f in enumerate(frames)
                if not (self.resume and _outputs_exist(self.output_path, self.folder, f.name, self.write_labels))]
        stats = {"frames": len(frames), "skipped_existing": len(frames) - len(todo), "written": 0, "inserted": 0}
        t_start = time.perf_counter()
        B = self.batch_size
        chunks = [todo[i:i + B] for i in range(0, len(todo), B)]
        read_q, errors, stop = queue.Queue(maxsize=2), [], threading.Event()
        pool = ThreadPoolExecutor(max_workers=max(1, int(io_threads)))

        base = os.path.join(self.output_path, self.folder)
        for sub in ("velodyne", "check") + (("labels",) if self.write_labels else ()):
            os.makedirs(os.path.join(base, sub), exist_ok=True)

        def look_one(i):
            # the frame's size from its file, its inserts from the caller: the points themselves are read by native
            # threads straight into the lane's pinned input (StreamedAugmenter.submit_files)
            return os.path.getsize(frames[i].velodyne_file) // 16, inserts_for(i)

        def reader():
            try:
                for chunk in chunks:
                    got = list(pool.map(look_one, chunk))
                    got += [got[-1]] * (B - len(chunk))             # the last batch: repeat its last frame, drop the copies
                    item = (chunk, [g[0] for g in got], [g[1] for g in got])
                    while not stop.is_set():                       # do not block for ever on a consumer that has failed
                        try:
                            read_q.put(item, timeout=0.5)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return
            except Exception as e:
                errors.append(e)
            finally:
                while not stop.is_set():
                    try:
                        read_q.put(None, timeout=0.5)
                        break
                    except queue.Full:
                        pass

        rt = threading.Thread(target=reader, daemon=True)
        rt.start()
        aug_box = [None]

        def consume(tag, results, accepted):
            chunk = tag
            aug = aug_box[0]
            names = [frames[i].name for i in chunk] + [None] * (B - len(chunk))
            path = lambda sub, ext: [None if n is None else os.path.join(base, sub, f"{n}.{ext}") for n in names]
            if label_2_for:                                        # object detection: label_2/{f}.txt first (OD tools/datasets.py:81-84)
                os.makedirs(os.path.join(base, "label_2"), exist_ok=True)
                # (create_annotation for the whole batch by native threads: 4 096 frames in a Python loop were 0.3 s of 0.6)
                what = [label_2_for(i, acc) for i, acc in zip(chunk, accepted)]
                n2 = len(what)
                src = (C.c_char_p * n2)(*[os.fsencode(w[0]) for w in what])
                dst = (C.c_char_p * n2)(*[os.fsencode(os.path.join(base, "label_2", f"{frames[i].name}.txt")) for i in chunk])
                extra = (C.c_char_p * n2)(*["".join(w[1]).encode() for w in what])
                from . import _lib
                _lib.check(_lib.load().r3d_host_append_text_files(src, dst, extra, n2, min(16, n2)), "r3d_host_append_text_files")
            # straight from the lane's buffers, by native threads: the lane is not submitted again before this returns
            aug.write_files(aug.current_lane, path("velodyne", "bin"), path("labels", "label") if self.write_labels else None,
                            path("check", "bin"))
            stats["written"] += len(chunk)
            stats["inserted"] += sum(1 for acc in accepted[:len(chunk)] for a in acc if a >= 0)

        def shape_of(item):
            chunk, sizes, ins = item
            K = max(len(x[0]) for x in ins)
            n_max = max(sizes)
            grow = max(sum(len(s) for s in x[0] if s is not None) for x in ins)
            srows = max((sum(len(x[0][k]) for x in ins if k < len(x[0]) and x[0][k] is not None) for k in range(K)), default=0)
            return K, n_max, grow, srows

        pending, caps, done = [None], (0, 0, 0, 0), [False]

        def segment():
            """Batches for the current lanes; stops (leaving the batch in `pending`) at one that needs larger lanes."""
            while not errors:
                item = pending[0] if pending[0] is not None else read_q.get()
                pending[0] = None
                if item is None:
                    done[0] = True
                    return
                K, n_max, grow, srows = shape_of(item)
                if K > caps[0] or n_max > caps[1] or grow > caps[2] or srows > caps[3]:
                    pending[0] = item
                    return
                chunk, sizes, ins = item
                fr = [frames[i] for i in chunk] + [frames[chunk[-1]]] * (B - len(chunk))
                yield (("files", [f.velodyne_file for f in fr], [f.label_file for f in fr]),
                       [x[0] for x in ins], [x[1] for x in ins], chunk)

        try:
            aug = None
            while not done[0] and not errors:
                if pending[0] is None:
                    pending[0] = read_q.get()
                    if pending[0] is None:
                        break
                K, n_max, grow, srows = shape_of(pending[0])       # (larger) lanes with 25 % headroom
                caps = (max(K, caps[0]), max(int(n_max * 1.25) + 64, caps[1]), max(int(grow * 1.25) + 64, caps[2]),
                        max(int(srows * 1.25) + 64, caps[3]))
                aug = aug_box[0] = None                            # free the old lanes first
                t_setup = time.perf_counter()
                aug = aug_box[0] = StreamedAugmenter(B, caps[1], caps[2], caps[0], caps[3], lanes=lanes, device=self.device,
                                                     check_cols=self.check_cols,
                                                     collapse_keep=-1 if self.road_label is None else self.road_label,
                                                     pack_threads=pack_threads, delta=delta)
                aug.merge_on_collect = False                       # (files only: written from the pinned input and the delta)
                stats["t_setup"] = stats.get("t_setup", 0.0) + time.perf_counter() - t_setup   # lanes: device + pinned memory
                aug.run(segment(), consume)
                for k, v in aug.times.items():
                    stats["t_" + k] = stats.get("t_" + k, 0.0) + v
        finally:
            stop.set()
            rt.join(timeout=30)
            pool.shutdown(wait=True)
        if errors:
            raise errors[0]
        stats["t_total"] = time.perf_counter() - t_start
        stats["frames_per_s"] = stats["written"] / stats["t_total"] if stats["t_total"] > 0 else 0.0
        return stats

    def run(self, frames, candidates_for, label_2_for=None, lanes=1):
        """Process every frame; returns a dict of counters and timings.  ``lanes`` > 1: that many batches are on the GPU
        at a time, each on a worker thread with its own HIP stream and device batch (the upload and host work of one
        batch overlap the kernels of another); results are written in frame order all the same.

        candidates_for(i) -> (slots, min_points): slots[k] = ordered list of M x 5 float64
        candidates of insert k of frame i, min_points[k] its acceptance threshold.
        label_2_for(i, accepted) -> (path of frame i's label_2 file, annotation lines of its inserted
        objects) for the object-detection flavour (OD tools/datasets.py:81-84; the lines come from
        ``Real3DAug.insertion.create_annotation_line``); accepted[k] = index of the accepted candidate or -1."""
        todo = [i for i, f in enumerate(frames)
                if not (self.resume and _outputs_exist(self.output_path, self.folder, f.name, self.write_labels, self.waymo))]
        stats = {"frames": len(frames), "skipped_existing": len(frames) - len(todo), "written": 0,
                 "inserted": 0, "t_read": 0.0, "t_process": 0.0, "t_write": 0.0}
        t_start = time.perf_counter()
        chunks = [todo[i:i + self.batch_size] for i in range(0, len(todo), self.batch_size)]
        read_q, write_q = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
        errors, stop = [], threading.Event()

        def put_unless_stopped(q, item):                        # a producer must not block for ever on a consumer that has failed
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.5)
                    return True
                except queue.Full:
                    pass
            return False

        def reader():
            try:
                for chunk in chunks:
                    t0 = time.perf_counter()
                    scenes = self._read_batch([frames[i] for i in chunk])
                    cands = [candidates_for(i) for i in chunk]
                    stats["t_read"] += time.perf_counter() - t0
                    if not put_unless_stopped(read_q, (chunk, scenes, cands)):
                        return
            except Exception as e:                              # surface in the main thread
                errors.append(e)
            finally:
                put_unless_stopped(read_q, None)

        def writer():
            try:
                while True:
                    item = write_q.get()
                    if item is None:
                        return
                    chunk, results, accepted = item
                    t0 = time.perf_counter()

                    def write_one(args):
                        i, (xyzi, label, check), acc = args
                        if self.waymo:
                            write_frame_waymo(self.output_path, self.folder, frames[i].name, xyzi, label)   # merged5, added5
                        else:
                            write_frame(self.output_path, self.folder, frames[i].name, xyzi, label, check,
                                        self.write_labels, label_2=label_2_for(i, acc) if label_2_for else None)

                    # the write and rename calls release the interpreter lock: the files of a batch go out side by side
                    list(write_pool.map(write_one, zip(chunk, results, accepted)))
                    stats["written"] += len(chunk)
                    stats["t_write"] += time.perf_counter() - t0
            except Exception as e:
                errors.append(e)

        from concurrent.futures import ThreadPoolExecutor
        write_pool = ThreadPoolExecutor(max_workers=8)
        threads = [threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)]
        for t in threads:
            t.start()
        def process_one(item):
            chunk, scenes, cands = item
            t0 = time.perf_counter()
            if lanes > 1:
                import torch
                st = getattr(self._lane_batches, "stream", None)
                if st is None:
                    st = self._lane_batches.stream = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(st):
                    results, accepted = self.process(scenes, [c[0] for c in cands], [c[1] for c in cands])
                    st.synchronize()
            else:
                results, accepted = self.process(scenes, [c[0] for c in cands], [c[1] for c in cands])
            return chunk, results, accepted, time.perf_counter() - t0

        def hand_over(done):
            chunk, results, accepted, secs = done
            stats["t_process"] += secs
            stats["inserted"] += sum(1 for a in accepted for x in a if x >= 0)
            while not errors:                                   # do not block on a dead writer
                try:
                    write_q.put((chunk, results, accepted), timeout=0.5)
                    break
                except queue.Full:
                    pass

        workers = ThreadPoolExecutor(max_workers=int(lanes)) if lanes > 1 else None
        pending = []                                            # futures in submission order
        try:
            while True:
                item = read_q.get()
                if item is None or errors:
                    break
                if workers is None:
                    hand_over(process_one(item))
                    continue
                pending.append(workers.submit(process_one, item))
                while pending and (len(pending) >= lanes or pending[0].done()):
                    hand_over(pending.pop(0).result())
            while pending and not errors:
                hand_over(pending.pop(0).result())
        except Exception as e:
            errors.append(e)
        finally:
            if workers is not None:
                workers.shutdown(wait=True)
        while threads[1].is_alive():
            try:
                write_q.put(None, timeout=0.5)
                break
            except queue.Full:
                if errors:
                    break
        threads[1].join(timeout=60)
        write_pool.shutdown(wait=True)
        stop.set()                                              # (the reader, if it is still waiting to hand over a batch)
        if errors:
            raise errors[0]
        stats["t_total"] = time.perf_counter() - t_start
        stats["frames_per_s"] = stats["written"] / stats["t_total"] if stats["t_total"] > 0 else 0.0
        return stats


def run_sharded_files(frames, inserts_for, output_path, folder, rank=None, world_size=None, device=None, dataset="semantic",
                      batch_size=64, lanes=None, label_2_for=None, resume=True):
    """BASELINE config C4 ("full sweep, scene-sharded across the GPUs of a node"): rank r of G takes frames r, r + G,
    r + 2G, ... and runs them file to file through its own GPU; no rank talks to another on the data path, every frame
    is written by exactly one rank (the reference shards by letting N copies of the script race for claim files,
    SS insertion.py:339-350; resume is by the existence of a frame's outputs, as there).

    inserts_for(i) -> (samples, min_points) of frame i: one placement per insert (``run_streamed``; the Waymo flavour,
    whose clouds are float64, goes through ``run(lanes=...)``: the same overlap of consecutive batches, whole clouds back).
    Returns this rank's counters; `frames` holds the rank's own frame indices."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    from .batch import shard_indices
    mine = shard_indices(len(frames), rank, world_size)
    pipe = AugmentPipeline(output_path, folder, dataset=dataset, batch_size=batch_size,
                           device=device or f"cuda:{rank}", resume=resume)
    local = [frames[i] for i in mine]
    if dataset == "waymo":
        # float64 frames (lidar/{f}.npy, SS tools/datasets.py:239-270): `lanes` batches in flight, each on its own thread,
        # stream and device batch (begin_f64 / results_f64); the float32 delta path below does not carry float64 clouds
        def cands64(j):
            smp, need = inserts_for(mine[j])
            return [[x] for x in smp], need
        st = pipe.run(local, cands64, lanes=lanes or 3)
        st.update(rank=rank, world_size=world_size, frame_indices=mine)
        return st
    st = pipe.run_streamed(local, lambda j: inserts_for(mine[j]), lanes=lanes,
                           label_2_for=(lambda j, acc: label_2_for(mine[j], acc)) if label_2_for else None)
    st.update(rank=rank, world_size=world_size, frame_indices=mine)
    return st
