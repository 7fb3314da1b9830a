#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own functions.

Run in the build container only (``/root/reference`` must exist; it is read-only and never
travels to the GPU box):

    python tests/golden/make_golden.py

The reference is imported from ``/root/reference/semantic_segmentation/Real3DAug`` with
bytecode writing disabled.  Its hot-path functions (``add_space_for_spherical``,
``fill_spherical``, ``geometrical_front_view``, ``smooth_out``, ``class_closing``) run
unmodified.  Two things around them are ours and are stated here so a reader can weigh them:

* scikit-image is not installed, so ``tools/closing.py:2-4`` cannot import.  The three symbols
  it uses are stood in (``img_as_ubyte``, ``rectangle``, ``closing`` = ``scipy.ndimage`` grey
  dilation then erosion, the routine scikit-image itself delegates to).  Fixtures therefore pin
  everything EXCEPT scikit-image's own closing; that call is "parity unpinned" (DESIGN.md).
* the visibility / cull / select block has no function in the reference (it is inline in
  ``__main__``, insertion.py:463-482, accept + concat :511-526); ``inline_block`` below re-states
  those statements around the imported functions, in the driver's order (:371-381, :449-461).

Only arrays are written (inputs, intermediates, outputs); no reference source is stored.
"""
import copy
import importlib
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/semantic_segmentation/Real3DAug"


def import_reference():
    from scipy import ndimage as ndi
    sk = types.ModuleType("skimage")
    sku = types.ModuleType("skimage.util")
    skm = types.ModuleType("skimage.morphology")
    sku.img_as_ubyte = lambda a: np.round(np.asarray(a) * 255).astype(np.uint8)
    skm.rectangle = lambda nrows, ncols: np.ones((nrows, ncols), dtype=np.uint8)
    skm.closing = lambda img, selem: ndi.grey_erosion(ndi.grey_dilation(img, footprint=selem),
                                                      footprint=selem)
    sk.util, sk.morphology = sku, skm
    sys.modules.update({"skimage": sk, "skimage.util": sku, "skimage.morphology": skm})
    sys.path.insert(0, REF)
    import insertion  # noqa: E402  (the reference module)
    return insertion


def inline_block(ref, scene_pcl, sample_pcl, scene_train, sample_train):
    """insertion.py:463-482 around the reference's arrays."""
    first = True
    visible_sample = np.array([])
    covered_scene = np.array([])
    idx = np.where(sample_train < scene_train)
    for k in range(len(idx[0])):
        pid = idx[0][k] * ref.NUMCOLUMN + idx[1][k]
        cov = scene_pcl[scene_pcl[:, 8] == pid]
        scene_pcl = scene_pcl[scene_pcl[:, 8] != pid]
        vis = sample_pcl[sample_pcl[:, 8] == pid]
        if first:
            first, visible_sample, covered_scene = False, vis, cov
        else:
            visible_sample = np.append(visible_sample, vis, axis=0)
            covered_scene = np.append(covered_scene, cov, axis=0)
    return scene_pcl, visible_sample, covered_scene, idx


def one_step(ref, scene9, sample5):
    """One insert step exactly as the driver sequences it; returns every intermediate."""
    R, C = ref.NUMROW, ref.NUMCOLUMN
    out = {}
    scene9, max_el, min_el = ref.fill_spherical(scene9)
    out["scene_sph"] = scene9[:, 3:6].copy()
    out["bounds"] = np.array([max_el, min_el])
    s_train, s_label, scene9 = ref.geometrical_front_view(scene9, R, C, max_el, min_el)
    out["scene_pix"] = scene9[:, 8].astype(np.int64)
    out["scene_train_raw"], out["scene_label_raw"] = s_train.copy(), s_label.copy()
    out["scene_closed_u8"] = ref.class_closing(s_label)
    s_train, s_label = ref.smooth_out(s_train, s_label)
    out["scene_train"], out["scene_label"] = s_train, s_label
    backup = copy.deepcopy(scene9)
    smp9 = ref.add_space_for_spherical(sample5)
    smp9, _, _ = ref.fill_spherical(smp9)
    out["sample_sph"] = smp9[:, 3:6].copy()
    m_train, m_label, smp9 = ref.geometrical_front_view(smp9, R, C, max_el, min_el, sample=True)
    out["sample_pix"] = smp9[:, 8].astype(np.int64)
    out["sample_train_raw"], out["sample_label_raw"] = m_train.copy(), m_label.copy()
    m_train, m_label = ref.smooth_out(m_train, m_label)
    out["sample_train"], out["sample_label"] = m_train, m_label
    scene_out, visible, covered, idx = inline_block(ref, copy.deepcopy(backup), smp9, s_train, m_train)
    out["vis_rows"], out["vis_cols"] = idx[0].astype(np.int32), idx[1].astype(np.int32)
    out["scene_out"] = scene_out
    out["visible_sample"] = visible.reshape(-1, 9)
    out["covered_scene"] = covered.reshape(-1, 9)
    return out, backup, smp9


def chain(ref, scene5, samples5, min_points):
    """K sequential inserts (one candidate each); returns merged cloud, added points, flags."""
    scene9 = ref.add_space_for_spherical(scene5)
    allvis = np.zeros((0, 9))
    accepted = []
    for smp5, need in zip(samples5, min_points):
        st, backup, _ = one_step(ref, scene9, smp5)
        vis = st["visible_sample"]
        if len(vis) == 0 or len(vis) < need:               # insertion.py:511-517
            scene9 = backup
            accepted.append(0)
            continue
        scene9 = np.append(st["scene_out"], vis, axis=0)   # :526
        allvis = np.append(allvis, vis, axis=0)
        accepted.append(1)
    return scene9, allvis, np.array(accepted, dtype=np.int32)


def row_index_map(rows9, base9):
    """Index of every row of rows9 inside base9 (rows are unique by construction of the tests)."""
    key = {r.tobytes(): i for i, r in enumerate(np.ascontiguousarray(base9[:, [0, 1, 2, 6, 7]]))}
    return np.array([key[r.tobytes()] for r in np.ascontiguousarray(rows9[:, [0, 1, 2, 6, 7]])],
                    dtype=np.int32)


def load_by_path(name, path):
    """A reference module that has no package-relative imports (tools/datasets.py), under its own name."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_save_bytes(flavour, merged9, allvis9, label_2_text="", additional_lines=()):
    """The files the REFERENCE's own save_data writes (SS tools/datasets.py:72-91, Waymo :287-303,
    OD tools/datasets.py:76-95), as bytes.  The dataset objects are made without their constructors
    (those walk a dataset directory and prompt); save_data only needs the attributes set here."""
    import tempfile
    ss = load_by_path("ref_ss_datasets", "/root/reference/semantic_segmentation/Real3DAug/tools/datasets.py")
    od = load_by_path("ref_od_datasets", "/root/reference/object_detection/Real3DAug/tools/datasets.py")
    merged9, allvis9 = np.array(merged9, dtype=np.float64, copy=True), np.array(allvis9, dtype=np.float64, copy=True)
    with tempfile.TemporaryDirectory() as tmp:
        rd = lambda *p: open(os.path.join(tmp, *p), "rb").read()
        if flavour == "semantic":
            ds = object.__new__(ss.SemanticKITTI)
            ds.config, ds.velodyne_list = {"path": {"output_path": tmp}}, np.array(["x"])
            for sub in ("velodyne", "labels", "check"):
                os.makedirs(os.path.join(tmp, "f", sub))
            ds.save_data(merged9, allvis9, "f", "000000", 0)
            return rd("f", "velodyne", "000000.bin"), rd("f", "labels", "000000.label"), rd("f", "check", "000000.bin")
        if flavour == "waymo":
            ds = object.__new__(ss.Waymo)
            ds.config, ds.velodyne_list, ds.LiDAR_location = {"path": {"output_path": tmp}}, np.array(["x"]), np.array([1.22, 0, 2])
            for sub in ("lidar", "labels_v3_2", "check"):
                os.makedirs(os.path.join(tmp, "f", sub))
            ds.save_data(merged9, allvis9, "f", "000000", 0)
            return rd("f", "lidar", "000000.npy"), rd("f", "labels_v3_2", "000000.npy"), rd("f", "check", "000000.npy")
        ds = object.__new__(od.KITTI)
        ds.data_path, ds.save_output_folder, ds.velodyne_list = os.path.join(tmp, "in"), tmp, np.array(["x"])
        os.makedirs(os.path.join(tmp, "in", "label_2"))
        with open(os.path.join(tmp, "in", "label_2", "000000.txt"), "w") as fh:
            fh.write(label_2_text)
        for sub in ("velodyne", "label_2", "check"):
            os.makedirs(os.path.join(tmp, "f", sub))
        ds.save_data(merged9, allvis9, "f", "000000", 0, list(additional_lines))
        return rd("f", "velodyne", "000000.bin"), rd("f", "check", "000000.bin"), rd("f", "label_2", "000000.txt")


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    ref = import_reference()
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("pcl-augmentation_amd.synth")

    # ---- small full-intermediate cases -------------------------------------------------------
    small = [
        dict(tag="s2k", seed=11, beams=8, naz=250, shuffle=False, kind="pedestrian", dist=8.0),
        dict(tag="s8k", seed=12, beams=16, naz=500, shuffle=True, kind="car", dist=7.0),
        dict(tag="s20k", seed=13, beams=32, naz=625, shuffle=False, kind="cyclist", dist=12.0),
    ]
    for cs in small:
        xyzi, label = synth.make_scene(cs["seed"], cs["beams"], cs["naz"], shuffle=cs["shuffle"])
        scene5 = synth.scene5_from_packed(xyzi, label)
        smp5 = synth.make_insert(cs["seed"] + 100, cs["kind"], centre_range=cs["dist"])
        st, _, _ = one_step(ref, ref.add_space_for_spherical(scene5), smp5)
        save(f"step_{cs['tag']}.npz", in_xyzi=xyzi, in_label=label, sample5=smp5,
             **{k: v for k, v in st.items()})

    # ---- edge cases (sample partly outside the elevation range, fully hidden, seam, far) -----
    xyzi, label = synth.make_scene(21, 16, 400)
    scene5 = synth.scene5_from_packed(xyzi, label)
    edge = {}
    tall = synth.make_insert(5, "pedestrian", centre_range=3.0)
    tall[:, 2] = tall[:, 2] * 3.0 + 2.0                         # sticks out above the top beam
    edge["above"] = tall
    low = synth.make_insert(6, "car", centre_range=2.5)
    low[:, 2] -= 1.0                                             # partly below the lowest beam
    edge["below"] = low
    hidden = synth.make_insert(7, "car", centre_range=55.0)      # behind the 40 m wall
    hidden[:, 2] += 2.0
    seam = synth.make_insert(8, "car", centre_range=6.0, centre_az=np.pi)   # straddles az = 0 | 2pi
    seam = np.vstack([seam, [[-6.0, 0.0, -0.5, 0.5, 10.0], [-6.0, -0.0, -0.4, 0.5, 10.0]]])
    edge["seam"] = seam
    for tag, smp5 in edge.items():
        st, _, _ = one_step(ref, ref.add_space_for_spherical(scene5), smp5)
        keep = ["bounds", "scene_pix", "sample_pix", "sample_sph", "sample_train", "sample_label",
                "sample_train_raw", "vis_rows", "vis_cols", "scene_out", "visible_sample",
                "covered_scene", "scene_train", "scene_label"]
        save(f"edge_{tag}.npz", in_xyzi=xyzi, in_label=label, sample5=smp5,
             **{k: st[k] for k in keep})

    # fully hidden sample: dense scene so that every pixel in front of it is occupied or closed
    hx, hl = synth.make_scene(22, 64, 1500)
    st, _, _ = one_step(ref, ref.add_space_for_spherical(synth.scene5_from_packed(hx, hl)), hidden)
    save("edge_hidden.npz", in_xyzi=hx, in_label=hl, sample5=hidden, **{k: st[k] for k in keep})

    # far pixel: every point of one pixel beyond 500 m (first hit overwrites the 500 init)
    far = scene5.copy()
    far[:40, 0:3] *= 150.0
    st, _, _ = one_step(ref, ref.add_space_for_spherical(far), edge["seam"])
    save("edge_far.npz", scene5=far, sample5=edge["seam"], bounds=st["bounds"],
         scene_pix=st["scene_pix"], scene_train_raw=st["scene_train_raw"],
         scene_label_raw=st["scene_label_raw"], scene_train=st["scene_train"],
         scene_label=st["scene_label"], scene_out=st["scene_out"],
         visible_sample=st["visible_sample"])

    # ---- K-insert chains (outer loop a10) with .bin/.label byte images -----------------------
    for tag, seed, beams, naz, kinds, od in [("c20k", 31, 32, 625, ["pedestrian", "cyclist", "car"], False),
                                             ("c8k_od", 32, 16, 500, ["car", "pedestrian", "cyclist", "car"], True)]:
        xyzi, label = synth.make_scene(seed, beams, naz, collapse_labels_to_road=od)
        scene5 = synth.scene5_from_packed(xyzi, label)
        samples = [synth.make_insert(seed * 50 + k, kind, rng_range=(5.0, 15.0)) for k, kind in enumerate(kinds)]
        samples.insert(1, synth.make_insert(99, "car", centre_range=55.0))   # a rejected candidate
        need = [20] * len(samples)
        need[1] = 5000                                        # ... rejected by the min_points test
        merged, allvis, acc = chain(ref, scene5, samples, need)
        if od:                                                # the reference's own save_data writes the bytes
            vb, cb, _ = ref_save_bytes("kitti", merged, allvis)
            lb = b""
        else:
            vb, lb, cb = ref_save_bytes("semantic", merged, allvis)
        save(f"chain_{tag}.npz", in_xyzi=xyzi, in_label=label,
             sample_sizes=np.array([len(s) for s in samples], dtype=np.int32),
             samples=np.vstack(samples), min_points=np.array(need, dtype=np.int32),
             merged=merged[:, [0, 1, 2, 6, 7]], merged_pix=merged[:, 8].astype(np.int64),
             all_visible=allvis[:, [0, 1, 2, 6, 7]], accepted=acc,
             velodyne_bin=np.frombuffer(vb, dtype=np.uint8), label_bin=np.frombuffer(lb, dtype=np.uint8),
             check_bin=np.frombuffer(cb, dtype=np.uint8))

    # ---- config C1: one 120k-point frame, one pedestrian (compact storage) -------------------
    xyzi, label = synth.make_scene(1)
    scene5 = synth.scene5_from_packed(xyzi, label)
    smp5 = synth.make_insert(1001, "pedestrian", centre_range=8.0)
    st, backup, smp9 = one_step(ref, ref.add_space_for_spherical(scene5), smp5)
    keep_idx = row_index_map(st["scene_out"], backup)
    vis_idx = row_index_map(st["visible_sample"], smp9)
    save("c1_120k.npz", scene_seed=np.array(1), sample5=smp5, bounds=st["bounds"],
         scene_pix=st["scene_pix"].astype(np.int32), sample_pix=st["sample_pix"].astype(np.int32),
         scene_train=st["scene_train"], scene_label=st["scene_label"].astype(np.int8),
         scene_train_raw=st["scene_train_raw"],
         sample_train=st["sample_train"], sample_label=st["sample_label"].astype(np.int8),
         vis_rows=st["vis_rows"], vis_cols=st["vis_cols"], keep_idx=keep_idx, visible_idx=vis_idx,
         scene_el=st["scene_sph"][:, 2], scene_az=st["scene_sph"][:, 1])


if __name__ == "__main__":
    main()
