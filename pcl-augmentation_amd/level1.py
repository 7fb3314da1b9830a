"""The reference's per-insert loop for ONE frame on the Level-1 kernels (whole range images in HBM, one launch per
function of ``Real3DAug/insertion.py``): what ``augment_batch`` falls back on, on the GPU, for a frame the batched
kernels turn down with ``R3D_S_WINDOW_TOO_LARGE`` -- an object so near the sensor on so fine a grid that its window of
the range image exceeds a CU's LDS even with every scratch image in the pool (a car 2-3 m away on 448 x 2880; never
on the reference's 112 x 1440).  Milliseconds per insert instead of microseconds, same bytes.

Follows SS Real3DAug/insertion.py:362 (scratch layout once), :371-381 (per insert: spherical fill, range image,
closing of the scene), :449-482 (candidates in order, each against the same scene), :511-526 (the first one whose
visible part reaches the threshold is appended), :534-545 (all_visible_parts).
"""
from __future__ import annotations

import numpy as np

from . import _lib


def augment_scene(xyzi, label, candidates, min_points, rows=_lib.NUMROW, cols=_lib.NUMCOLUMN, device="cuda:0", check_cols=5):
    """One frame: (xyzi float32 [n,4], label uint32 [n]) and candidates[k] = ordered list of M x 5 float64 placements
    of insert k.  Returns ((xyzi float32 [n',4], label uint32 [n'], check float32 [m,check_cols] or None), accepted)
    with accepted[k] = index of the accepted candidate or -1 -- what ``augment_batch`` returns for the frame."""
    torch = _lib.require_gpu()
    from .Real3DAug import insertion as ins
    from .Real3DAug.tools import closing
    from .Real3DAug.tools.datasets import pack_for_save
    # the grid is the reference's two globals (insertion.py:22-23); they are NOT edited here (another thread may be inside
    # the Level-1 mirrors): the multiplier of the pixel ids goes to the private helpers as an argument
    return _augment_scene(torch, ins, closing, pack_for_save, xyzi, label, candidates, min_points, int(rows), int(cols), device, check_cols)


def _augment_scene(torch, ins, closing, pack_for_save, xyzi, label, candidates, min_points, rows, cols, device, check_cols):
    with _lib.on(device):
        s5 = np.hstack((np.asarray(xyzi, dtype=np.float32).astype(np.float64),
                        (np.asarray(label).astype(np.uint32) & 0xFFFF).astype(np.float64)[:, None]))
        scene = ins.add_space_for_spherical(torch.from_numpy(s5).to(device))
        visible_parts, accepted = [], []
        for cands, need in zip(candidates, min_points):
            chosen = -1
            if any(c is not None and len(c) for c in cands):
                scene, max_el, min_el = ins.fill_spherical(scene)
                train, lab, scene = ins._front_view(scene, rows, cols, max_el, min_el, False, cols)
                train, lab = closing.smooth_out(train, lab)
            for ci, smp in enumerate(cands):
                if smp is None or not len(smp):
                    continue
                sm9 = ins.add_space_for_spherical(torch.from_numpy(np.ascontiguousarray(smp, dtype=np.float64)).to(device))
                sm9, _, _ = ins.fill_spherical(sm9)
                s_train, s_lab, sm9 = ins._front_view(sm9, rows, cols, max_el, min_el, True, cols)
                s_train, s_lab = closing.smooth_out(s_train, s_lab)
                out, vis, _ = ins._merge(scene, sm9, train, s_train, cols)
                if len(vis) == 0 or len(vis) < need:              # :511-517
                    continue
                scene = torch.cat((out, vis), dim=0)              # :526
                visible_parts.append(vis)
                chosen = ci
                break
            accepted.append(chosen)
        out_xyzi, out_label, _ = pack_for_save(scene, 5)
        check = None
        if check_cols:
            added = torch.cat(visible_parts, dim=0) if visible_parts else torch.zeros((0, 9), dtype=torch.float64, device=device)
            check = pack_for_save(added, check_cols)[2]
    return (out_xyzi, out_label, check), accepted
