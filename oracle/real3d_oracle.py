"""CPU oracle for the Real3D-Aug occlusion-handling + insertion-merge hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``pcl-augmentation_amd``) never
routes through this file and fails loudly when its HIP library is missing.

What it is: a NumPy restatement of the reference algorithm (ctu-vras/pcl-augmentation,
``semantic_segmentation/Real3DAug/insertion.py`` + ``tools/closing.py``; the
``object_detection`` copies are byte-identical one line lower).  Every function cites the
reference lines it follows.  Two forms are kept on purpose:

* ``*_loop`` functions follow the reference statement by statement (pure-Python loops, small
  inputs only).  They pin the vectorised forms.
* the un-suffixed functions are vectorised and finish a 120k-point scene in well under a
  second; they are what the GPU parity tests and the CPU baseline use.

Pinning: the reference ships no tests or golden vectors, so the oracle is
pinned by outputs of the reference's own functions, imported in the build container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks ``array_equal`` on every intermediate).  One call on
the path leaves the reference tree: ``skimage.morphology.closing`` (closing.py:2-4,19-21;
scikit-image version unpinned upstream and not installed here).  The golden generator stands
it in with ``scipy.ndimage`` grey dilation→erosion, which is what scikit-image delegates to;
at that single call parity is therefore "unpinned" against a real scikit-image build and
pinned only against the mathematical definition (binary closing, 5x3 all-ones footprint).
"""
from __future__ import annotations

import math

import numpy as np

NUMROW = 112        # insertion.py:22
NUMCOLUMN = 360 * 4  # insertion.py:23
EMPTY_DEPTH = 500.0  # insertion.py:99
TWO_PI = 2 * math.pi


# ----------------------------------------------------------------------------------------------
# a1  add_space_for_spherical                                             insertion.py:54-64
# ----------------------------------------------------------------------------------------------
def add_space_for_spherical(point_cloud: np.ndarray) -> np.ndarray:
    """N x 5 (x y z intensity label) -> N x 9 scratch record, -1 filled (insertion.py:60-64)."""
    n = len(point_cloud)
    out = np.ones((n, 9)) * -1
    if n:
        out[:, 0:3] = point_cloud[:, 0:3]
        out[:, 6:8] = point_cloud[:, 3:5]
    return out


# ----------------------------------------------------------------------------------------------
# a2  fill_spherical                                                      insertion.py:67-81
# ----------------------------------------------------------------------------------------------
def fill_spherical(point_cloud: np.ndarray):
    """In place: r, azimuth (+pi), elevation = acos(z/r); returns (pcl, max_el, min_el).

    insertion.py:74-76 for the three columns, :78-79 for the bounds.  ``x ** 2`` is evaluated
    by NumPy as ``x * x`` and the adds run left to right, which is what is written here.
    """
    x = point_cloud[:, 0]
    y = point_cloud[:, 1]
    z = point_cloud[:, 2]
    point_cloud[:, 3] = np.sqrt(x * x + y * y + z * z)
    point_cloud[:, 4] = np.arctan2(y, x) + np.pi
    point_cloud[:, 5] = np.arccos(z / point_cloud[:, 3])
    min_el = np.min(point_cloud[:, 5])
    max_el = np.max(point_cloud[:, 5])
    return point_cloud, max_el, min_el


# ----------------------------------------------------------------------------------------------
# a3  geometrical_front_view                                              insertion.py:84-129
# ----------------------------------------------------------------------------------------------
def bin_rows_cols(point_cloud, num_row, num_column, max_el, min_el):
    """Row / column of every point, truncated toward zero like ``int()`` (insertion.py:104-105).

    Returned as float64 holding integers (or +-inf / nan where ``int()`` would have raised) so
    that callers can apply the reference's range tests without overflow.
    """
    d_el = (max_el - min_el) / num_row                    # :96
    d_az = 2 * math.pi / num_column                       # :97
    with np.errstate(all="ignore"):
        row = np.trunc((point_cloud[:, 5] - min_el - 0.00001) / d_el)        # :104
        col = np.trunc(np.remainder(point_cloud[:, 4], 2 * math.pi) / d_az)  # :105
    return row, col


def geometrical_front_view(point_cloud, num_row, num_column, max_el, min_el, sample=False):
    """Vectorised range image: (train, label, point_cloud) exactly as insertion.py:84-129.

    * train[row, col] = min r over the points of the pixel, 500 where empty.  The reference's
      first hit overwrites the 500 unconditionally (:122-125) and later hits only lower it
      (:118-120), so the result is the plain minimum even when every r exceeds 500.
    * label[row, col] = 1 where any point landed, -1 elsewhere (:98, :123).
    * column 8 = row * NUMCOLUMN + col with the GLOBAL constant (:116, :127); with
      ``sample=True`` points whose row is outside [0, num_row) are skipped and keep their
      previous column-8 value (:107-108); otherwise out-of-range raises like the asserts
      (:110-112).
    """
    n = len(point_cloud)
    label = np.ones((num_row, num_column)) * -1
    train = np.ones((num_row, num_column)) * EMPTY_DEPTH
    if n == 0:
        return train, label, point_cloud
    row, col = bin_rows_cols(point_cloud, num_row, num_column, max_el, min_el)
    row_ok = (row >= 0) & (row < num_row)
    if sample:
        use = row_ok
    else:
        if not np.all(row_ok):
            raise AssertionError("Rows in FoV went something wrong.")
        use = np.ones(n, dtype=bool)
    if not np.all((col[use] >= 0) & (col[use] < num_column)):
        raise AssertionError("Column in FoV went something wrong.")
    r_i = row[use].astype(np.int64)
    c_i = col[use].astype(np.int64)
    flat = r_i * num_column + c_i
    tr = train.reshape(-1)
    first = np.full(num_row * num_column, np.inf)
    np.minimum.at(first, flat, point_cloud[use, 3])
    hit = np.zeros(num_row * num_column, dtype=bool)
    hit[flat] = True
    tr[hit] = first[hit]
    label.reshape(-1)[hit] = 1
    point_cloud[use, 8] = r_i * NUMCOLUMN + c_i
    return train, label, point_cloud


def geometrical_front_view_loop(point_cloud, num_row, num_column, max_el, min_el, sample=False):
    """Statement-by-statement form of insertion.py:94-129 (small inputs; pins the vector form)."""
    d_el = (max_el - min_el) / num_row
    d_az = 2 * math.pi / num_column
    label = np.ones((num_row, num_column)) * -1
    train = np.ones((num_row, num_column)) * EMPTY_DEPTH
    for i in range(len(point_cloud)):
        p = point_cloud[i]
        prow = int((p[5] - min_el - 0.00001) / d_el)
        pcol = int((p[4] % (2 * math.pi)) / d_az)
        if sample and not (num_row > prow >= 0):
            continue
        assert num_row > prow >= 0
        assert num_column > pcol >= 0
        if label[prow][pcol] != -1:
            point_cloud[i][8] = prow * NUMCOLUMN + pcol
            if train[prow][pcol] > p[3]:
                train[prow][pcol] = p[3]
        else:
            label[prow][pcol] = 1
            train[prow][pcol] = p[3]
            point_cloud[i][8] = prow * NUMCOLUMN + pcol
    return train, label, point_cloud


# ----------------------------------------------------------------------------------------------
# a4  class_closing                                                        closing.py:9-23
# ----------------------------------------------------------------------------------------------
def _shift_or(a: np.ndarray, fill: bool):
    """5-row x 3-column window reduction with border clipping; OR when fill=False, AND when True.

    ``rectangle(5, 3)`` is 5 rows x 3 columns (closing.py:20).  Grey dilation of a {0,255} image
    is OR over the window, erosion is AND; scikit-image's reflect border equals clipping the
    window for an odd symmetric footprint, i.e. out-of-image taps are neutral.
    """
    rows, cols = a.shape
    out = a.copy()
    for dr in range(-2, 3):
        for dc in range(-1, 2):
            if dr == 0 and dc == 0:
                continue
            sh = np.full_like(a, fill)
            r0, r1 = max(0, -dr), min(rows, rows - dr)
            c0, c1 = max(0, -dc), min(cols, cols - dc)
            sh[r0:r1, c0:c1] = a[r0 + dr:r1 + dr, c0 + dc:c1 + dc]
            out = (out & sh) if fill else (out | sh)
    return out


def class_closing(original_label: np.ndarray) -> np.ndarray:
    """uint8 {0,255}: closing of clip(label,0,1) with a 5x3 all-ones element (closing.py:15-21)."""
    occ = np.clip(original_label, 0, 1) >= 0.5          # img_as_ubyte rounds; inputs are {0,1}
    dil = _shift_or(occ, fill=False)
    clo = _shift_or(dil, fill=True)
    return np.where(clo, 255, 0).astype(np.uint8)


# ----------------------------------------------------------------------------------------------
# a5  smooth_out                                                           closing.py:26-62
# ----------------------------------------------------------------------------------------------
def smooth_out(original_train: np.ndarray, original_label: np.ndarray):
    """Hole fill: closed-but-empty pixels get the mean of the occupied 5x3 neighbours.

    closing.py:38-59.  The sum runs drow = -2..2 outer, dcolumn = -1..1 inner over ORIGINAL
    depths (:46-51) and is divided by the neighbour count (:57); adding 0.0 for a skipped tap
    leaves an IEEE double sum unchanged, so the masked vector sum below is bit-identical.
    The ``neighbors == 0`` branch (:52-55) cannot be reached: closing is a subset of dilation.
    """
    train = original_train.copy()
    label = original_label.copy()
    closed = class_closing(original_label)
    rows, cols = original_label.shape
    occ = original_label == 1
    hole = (closed == 255) & ~occ                         # :40-42 negated
    total = np.zeros((rows, cols))
    count = np.zeros((rows, cols), dtype=np.int64)
    for dr in range(-2, 3):
        for dc in range(-1, 2):
            r0, r1 = max(0, -dr), min(rows, rows - dr)
            c0, c1 = max(0, -dc), min(cols, cols - dc)
            o = np.zeros((rows, cols), dtype=bool)
            v = np.zeros((rows, cols))
            o[r0:r1, c0:c1] = occ[r0 + dr:r1 + dr, c0 + dc:c1 + dc]
            v[r0:r1, c0:c1] = original_train[r0 + dr:r1 + dr, c0 + dc:c1 + dc]
            total = total + np.where(o, v, 0.0)
            count = count + o
    fill = hole & (count > 0)
    train[fill] = total[fill] / count[fill]
    label[hole] = 1
    return train, label


def smooth_out_loop(original_train, original_label):
    """Statement-by-statement form of closing.py:34-62 (small grids; pins the vector form)."""
    train = original_train.copy()
    label = original_label.copy()
    closed = class_closing(original_label)
    nr, nc = original_label.shape
    for row in range(nr):
        for column in range(nc):
            if (closed[row][column] == 255 and label[row][column] == 1) or closed[row][column] == 0:
                continue
            neighbors = 0
            sum_distance = 0
            for drow in range(-2, 3):
                for dcolumn in range(-1, 2):
                    if -1 < drow + row < nr and -1 < dcolumn + column < nc \
                            and original_label[drow + row][dcolumn + column] == 1:
                        neighbors += 1
                        sum_distance += original_train[row + drow][column + dcolumn]
            if neighbors == 0:
                label[row][column] = 1
            else:
                train[row][column] = sum_distance / neighbors
                label[row][column] = 1
    return train, label


# ----------------------------------------------------------------------------------------------
# a6-a8  visibility mask, cull + select, concat                   insertion.py:463-482, 511-526
# ----------------------------------------------------------------------------------------------
def occlusion_merge(scene_pcl, sample_pcl, scene_train, sample_train):
    """The unnamed inline block: returns (scene_out, visible_sample, covered_scene).

    * visible pixels = ``np.where(sample_train < scene_train)`` in row-major order (:467);
    * scene_out = scene rows whose column 8 is not a visible pixel, original order (:472-473);
    * visible_sample = sample rows whose column 8 is a visible pixel, grouped by pixel in
      row-major pixel order, original sample order inside a pixel (:474-482);
    * covered_scene = removed scene rows in the same grouping (:470-471, :479-482).
    Pixel ids use NUMCOLUMN (:470), so grids must be NUMROW x NUMCOLUMN like the reference's.
    With no visible pixel the reference leaves ``np.array([])`` in both lists (:463-464).
    """
    vis = sample_train < scene_train
    if not vis.any():
        return scene_pcl, np.array([]), np.array([])
    rr, cc = np.nonzero(vis)
    vis_ids = rr * NUMCOLUMN + cc                        # ascending (row-major)
    s_pix = scene_pcl[:, 8]
    m_pix = sample_pcl[:, 8]
    s_hit = np.isin(s_pix, vis_ids)
    m_hit = np.isin(m_pix, vis_ids)
    scene_out = scene_pcl[~s_hit]
    cov = scene_pcl[s_hit]
    covered_scene = cov[np.argsort(cov[:, 8], kind="stable")]
    v = sample_pcl[m_hit]
    visible_sample = v[np.argsort(v[:, 8], kind="stable")]
    return scene_out, visible_sample, covered_scene


def occlusion_merge_loop(scene_pcl, sample_pcl, scene_train, sample_train):
    """Statement-by-statement form of insertion.py:463-482 (small inputs; pins the vector form)."""
    first_part = True
    visible_sample = np.array([])
    covered_scene = np.array([])
    indexes = np.where(sample_train < scene_train)
    for ind in range(len(indexes[0])):
        pid = indexes[0][ind] * NUMCOLUMN + indexes[1][ind]
        covered_part = scene_pcl[scene_pcl[:, 8] == pid]
        scene_pcl = scene_pcl[scene_pcl[:, 8] != pid]
        visible_part = sample_pcl[sample_pcl[:, 8] == pid]
        if first_part:
            first_part = False
            visible_sample = visible_part
            covered_scene = covered_part
        else:
            visible_sample = np.append(visible_sample, visible_part, axis=0)
            covered_scene = np.append(covered_scene, covered_part, axis=0)
    return scene_pcl, visible_sample, covered_scene


# ----------------------------------------------------------------------------------------------
# a9  remove_space_for_spherical / save_data byte images       SS datasets.py:72-106, OD :76-109
# ----------------------------------------------------------------------------------------------
def remove_space_for_spherical(point_cloud):
    """(N x 4 xyz+intensity, N x 1 label) float64, SS datasets.py:93-106."""
    n = len(point_cloud)
    labels = np.zeros((n, 1))
    pcl = np.ones((n, 4)) * -1
    if n:
        pcl[:, 0:3] = point_cloud[:, 0:3]
        pcl[:, 3] = point_cloud[:, 6]
        labels[:, 0] = point_cloud[:, 7]
    return pcl, labels


def save_bytes_semantic(point_cloud, added_points):
    """Bytes of velodyne/{f}.bin, labels/{f}.label, check/{f}.bin (SS datasets.py:72-89)."""
    pc, lab = remove_space_for_spherical(point_cloud)
    ap, al = remove_space_for_spherical(added_points)
    ap = np.hstack((ap, al))
    return (pc.astype(np.float32).tobytes(), lab.astype(np.uint32).tobytes(),
            ap.astype(np.float32).tobytes())


def save_bytes_kitti(point_cloud, added_points):
    """Bytes of velodyne/{f}.bin and check/{f}.bin (OD datasets.py:76-93; labels not written)."""
    pc, _ = remove_space_for_spherical(point_cloud)
    ap, _ = remove_space_for_spherical(added_points)
    return pc.astype(np.float32).tobytes(), ap.astype(np.float32).tobytes()


# ----------------------------------------------------------------------------------------------
# a10  per-insert outer loop, with the accept test               insertion.py:371-381, 449-545
# ----------------------------------------------------------------------------------------------
def scene_field_of_view(scene_pcl9, num_row=NUMROW, num_column=NUMCOLUMN):
    """Lines :373-377: spherical fill, range image, closing.  Returns grids and the bounds."""
    scene_pcl9, max_el, min_el = fill_spherical(scene_pcl9)
    train, label, scene_pcl9 = geometrical_front_view(scene_pcl9, num_row, num_column, max_el, min_el)
    train, label = smooth_out(train, label)
    return scene_pcl9, train, label, max_el, min_el


def evaluate_candidate(scene_pcl9, scene_train, max_el, min_el, sample5,
                       num_row=NUMROW, num_column=NUMCOLUMN):
    """Lines :455-482 for one placement candidate (the scene copy of :453 is the caller's)."""
    sample9 = add_space_for_spherical(sample5)
    sample9, _, _ = fill_spherical(sample9)
    s_train, s_label, sample9 = geometrical_front_view(sample9, num_row, num_column, max_el, min_el,
                                                        sample=True)
    s_train, s_label = smooth_out(s_train, s_label)
    return occlusion_merge(scene_pcl9, sample9, scene_train, s_train)


def augment_scene(scene5, candidates_per_insert, min_points, num_row=None, num_column=None):
    """Run the K-insert chain of one frame.

    The grid is the module's NUMROW x NUMCOLUMN, read at call time: a user of the reference changes
    the range-image size by editing those two globals (insertion.py:22-23), which also moves the
    pixel ids of column 8 -- tests of other grid sizes patch both globals and nothing else.

    ``candidates_per_insert[k]`` is the ordered list of placement candidates (M x 5 float64
    arrays) tried for insert k, ``min_points[k]`` its acceptance threshold.  Follows
    insertion.py:362 (scratch layout once), :371-381 (per insert: refresh the scene field of
    view), :449-482 (evaluate candidates in order on a copy of the scene), :511-526 (accept the
    first candidate with ``len(visible_sample) >= min_points`` and append it), :534-545
    (``all_visible_parts``).  Returns (scene_pcl9, all_visible_parts9, accepted_index_per_insert).
    An insert whose candidates all fail leaves the scene unchanged and records -1.
    """
    num_row = NUMROW if num_row is None else num_row
    num_column = NUMCOLUMN if num_column is None else num_column
    scene = add_space_for_spherical(np.asarray(scene5, dtype=np.float64))
    all_visible = np.zeros((0, 9))
    accepted = []
    for cands, need in zip(candidates_per_insert, min_points):
        scene, s_train, _, max_el, min_el = scene_field_of_view(scene, num_row, num_column)
        backup = scene
        chosen = -1
        for ci, sample5 in enumerate(cands):
            out, visible, _ = evaluate_candidate(backup, s_train, max_el, min_el,
                                                 np.asarray(sample5, dtype=np.float64), num_row, num_column)
            if len(visible) == 0 or len(visible) < need:       # :511-517
                continue
            scene = np.append(out, visible, axis=0)            # :526
            all_visible = np.append(all_visible, visible, axis=0)
            chosen = ci
            break
        accepted.append(chosen)
    return scene, all_visible, accepted
