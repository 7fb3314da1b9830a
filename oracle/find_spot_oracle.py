"""CPU restatement of the reference's placement search (TEST INFRASTRUCTURE ONLY).

SURVEY.md par.8 row f-1: ``find_possible_places`` of
semantic_segmentation/Real3DAug/tools/find_spot.py:192-273 with its helpers
``rotate_bounding_box_2`` (:42-76), ``check_bounding_box`` (:79-104), ``correct_height``
(:107-152), ``read_label_line`` (:155-189) and ``cut_bounding_box`` (tools/cut_bbox.py:7-68).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline leg may import this
file; the product path (``pcl-augmentation_amd``) never does.

Third-party arithmetic the reference calls on this path and that is not under /root/reference
(both unpinned there: the reference ships no requirements file):

* ``scipy.spatial.transform.Rotation`` -- ``from_quat``, ``as_dcm`` (today ``as_matrix``),
  ``from_dcm`` (``from_matrix``), ``as_quat``.  Restated below (`quat_normalize`,
  `quat_to_matrix`, `matrix_to_quat`) after SciPy 1.15.3, the version in this image, and checked
  bit for bit against it in tests/test_find_spot_oracle.py.
* BLAS behind ``np.dot`` / ``@`` for the 3x3 and 4x4 products (OpenBLAS 0.3.29 in this image).
  Its kernels accumulate a dot product with fused multiply-adds in a fixed order:
  matrix x matrix  ``acc = a0*b0; acc = fma(a1, b1, acc); acc = fma(a2, b2, acc) ...``
  3x3 x column     ``fma(a2, b2, fma(a0, b0, a1*b1))``
  4x4 x column     ``(a0*b0 + a2*b2) + (a1*b1 + a3*b3)``, every product and sum rounded
  (the two column forms are also what ``@`` does at :72 and :236 for a sample of ONE point)
  (measured here; a BLAS without FMA gives results 1 ULP away for ~3 % of the points).  The
  restatement reproduces exactly that with an emulated FMA, so it does not depend on the BLAS of
  the machine the tests run on.

Pinned by tests/golden/places_*.npz, written by tests/golden/make_golden_places.py from the
reference's own ``find_possible_places`` run in this container.
"""
import math

import numpy as np

DEG1 = np.deg2rad(1)                         # find_spot.py:52 with the default rotation=1
COS1, SIN1 = float(np.cos(DEG1)), float(np.sin(DEG1))   # :57-59


# ---- exact fused multiply-add on float64 arrays ---------------------------------------------------
def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _split(a):
    c = 134217729.0 * a
    hi = c - (c - a)
    return hi, a - hi


def _two_prod(a, b):
    p = a * b
    ah, al = _split(a)
    bh, bl = _split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def fma(a, b, c):
    """round(a*b + c) with one rounding (Boldo & Melquiond: error-free product and sum, the small
    terms added with rounding to odd).  Valid away from overflow / underflow."""
    a, b, c = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                                  np.asarray(c, dtype=np.float64))
    p, e = _two_prod(a, b)
    s, t = _two_sum(p, c)                      # a*b + c == s + t + e exactly
    uh, ul = _two_sum(t, e)
    uh = np.array(uh, copy=True, ndmin=1)
    ul = np.asarray(ul).reshape(uh.shape)
    iv = uh.view(np.int64)
    adj = (ul != 0) & ((iv & 1) == 0)
    away = (ul > 0) == (uh > 0)
    iv[adj & away] += 1
    iv[adj & ~away] -= 1
    return (np.asarray(s).reshape(uh.shape) + uh).reshape(np.shape(s))


def blas_matmul(A, B):
    """A (m,k) @ B (k,n) the way the BLAS gemm kernels accumulate (see the module docstring)."""
    acc = A[:, 0][:, None] * B[0][None, :]
    for k in range(1, A.shape[1]):
        acc = fma(A[:, k][:, None], B[k][None, :], acc)
    return acc


def blas_matvec3(A, v):
    """np.dot(A (3,3), v (3,1)) the way the BLAS gemv kernel accumulates."""
    v = np.asarray(v, dtype=np.float64).reshape(3)
    return fma(A[:, 2], v[2], fma(A[:, 0], v[0], A[:, 1] * v[1]))


def blas_matvec4(A, v):
    """A (4,4) @ v (4,1): the gemv kernel's row dot products, two partial sums (even and odd terms)
    of rounded products added at the end."""
    v = np.asarray(v, dtype=np.float64).reshape(4)
    return (A[:, 0] * v[0] + A[:, 2] * v[2]) + (A[:, 1] * v[1] + A[:, 3] * v[3])


# ---- scipy.spatial.transform.Rotation, the four calls on the path --------------------------------
def quat_normalize(q):
    """Rotation.from_quat: the stored quaternion is q / |q| (scalar-last)."""
    x, y, z, w = (float(v) for v in q)
    n = math.sqrt(x * x + y * y + z * z + w * w)
    return np.array([x / n, y / n, z / n, w / n])


def quat_to_matrix(q):
    """Rotation.as_matrix (as_dcm in the SciPy the reference was written for) of a unit quaternion."""
    x, y, z, w = (float(v) for v in q)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return np.array([[x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)],
                     [2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)],
                     [2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2]])


def matrix_to_quat(m):
    """Rotation.from_matrix(m).as_quat(): largest of (diagonal, trace) picks the branch."""
    d = [m[0, 0], m[1, 1], m[2, 2], m[0, 0] + m[1, 1] + m[2, 2]]
    c = int(np.argmax(d))
    q = [0.0] * 4
    if c != 3:
        i = c
        j = (i + 1) % 3
        k = (j + 1) % 3
        q[i] = 1 - d[3] + 2 * m[i, i]
        q[j] = m[j, i] + m[i, j]
        q[k] = m[k, i] + m[i, k]
        q[3] = m[k, j] - m[j, k]
    else:
        q[0] = m[2, 1] - m[1, 2]
        q[1] = m[0, 2] - m[2, 0]
        q[2] = m[1, 0] - m[0, 1]
        q[3] = 1 + d[3]
    q = [float(v) for v in q]
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    return np.array([v / n for v in q])


# ---- annotations -----------------------------------------------------------------------------------
def make_annotation(center, quat, length, width, height, cls=None):
    """The reference's annotation dictionary (find_spot.py:15-27)."""
    return {"center": {"x": center[0], "y": center[1], "z": center[2]},
            "rotation": {"x": quat[0], "y": quat[1], "z": quat[2], "w": quat[3]},
            "length": length, "width": width, "height": height, "class": cls}


def read_label_line(line):
    """find_spot.py:155-189: 'class x y z height length width rot_z' -> annotation."""
    it = line.split(" ")
    x, y, z = float(it[1]), float(it[2]), float(it[3])
    height, width, length = float(it[4]), float(it[6]), float(it[5])
    a = float(it[7])
    m = np.array([[math.cos(a), -1 * math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], dtype=np.float64)
    q = matrix_to_quat(m)                                                  # :177-179
    # :181-183: the dictionary's 'length' is the label's width column and vice versa
    return make_annotation([x, y, z], q, width, length, height, [it[0]])


def anno_center(a):
    return np.array([a["center"]["x"], a["center"]["y"], a["center"]["z"]], dtype=np.float64)


def anno_quat(a):
    return np.array([a["rotation"]["x"], a["rotation"]["y"], a["rotation"]["z"], a["rotation"]["w"]], dtype=np.float64)


Z1 = np.array([[COS1, -SIN1, 0.0], [SIN1, COS1, 0.0], [0.0, 0.0, 1.0]])      # :57-59


def rotate_bounding_box_2(bbox_pcl, annotation):
    """find_spot.py:42-76 for rotation=1: one more degree about the sensor's z axis, applied to
    the points in place, to the box centre and (from the right) to the box orientation."""
    rot = quat_to_matrix(quat_normalize(anno_quat(annotation)))             # :53-55
    q = matrix_to_quat(blas_matmul(rot, Z1))                                # :61-65
    c = blas_matvec3(Z1, anno_center(annotation))                           # :66-70
    if len(bbox_pcl) == 1:                                                  # :72 with one column: matrix x vector
        bbox_pcl[0, :3] = blas_matvec3(Z1, bbox_pcl[0, :3])
    else:
        bbox_pcl[:, :3] = blas_matmul(Z1, bbox_pcl[:, :3].T).T              # :72
    return bbox_pcl, make_annotation(c, q, annotation["length"], annotation["width"], annotation["height"],
                                     annotation["class"])


def box_mask(points_xyz, annotation):
    """tools/cut_bbox.py:7-68 as a mask: strictly inside the six faces; the box spans
    [-l/2, l/2] x [-w/2, w/2] x [0, h] in its own frame (the centre is the bottom centre)."""
    xc, yc, zc = (float(v) for v in anno_center(annotation))
    R = quat_to_matrix(quat_normalize(anno_quat(annotation)))               # :26-28
    x, y, z = points_xyz[:, 0], points_xyz[:, 1], points_xyz[:, 2]
    ext = [(annotation["length"], annotation["length"]), (annotation["width"], annotation["width"]),
           (annotation["height"], 0)]
    keep = np.ones(len(points_xyz), dtype=bool)
    for a in range(3):
        r0, r1, r2 = R[0][a], R[1][a], R[2][a]
        lhs = r0 * x + r1 * y + r2 * z
        hi, lo = ext[a]
        if a < 2:                                                           # :30-52: +- size / 2
            up = r0 * (xc + r0 * hi / 2) + r1 * (yc + r1 * hi / 2) + r2 * (zc + r2 * hi / 2)
            dn = r0 * (xc - r0 * lo / 2) + r1 * (yc - r1 * lo / 2) + r2 * (zc - r2 * lo / 2)
        else:                                                               # :54-64: 0 .. height
            up = r0 * (xc + r0 * hi) + r1 * (yc + r1 * hi) + r2 * (zc + r2 * hi)
            dn = r0 * (xc - r0 * 0) + r1 * (yc - r1 * 0) + r2 * (zc - r2 * 0)
        keep &= (lhs < up) & (lhs > dn)
    return keep


def cut_bounding_box(point_cloud, annotation):
    return point_cloud[box_mask(point_cloud, annotation)]


def check_bounding_box(scene_pcl, scene_anno, sample_pcl, sample_anno, ok_surface):
    """find_spot.py:79-104: no scene point other than placement surface inside the sample's box,
    and no sample point inside any annotated scene box."""
    inside = scene_pcl[box_mask(scene_pcl, sample_anno)]
    if len(inside) and not np.isin(inside[:, 7], ok_surface).all():        # :92-96
        return False
    for a in scene_anno:                                                    # :99-103
        if box_mask(sample_pcl, a).any():
            return False
    return True


def correct_height(scene_pcl, sample_pcl, sample_anno, ok_surface):
    """find_spot.py:107-152: growing-radius search (0.1 m steps) around the box centre for
    placement-surface points above z = -3; the box is put on their mean height."""
    cx, cy, cz = (float(v) for v in anno_center(sample_anno))
    d2 = (scene_pcl[:, 0] - cx) ** 2 + (scene_pcl[:, 1] - cy) ** 2         # :123
    surface = np.zeros((0, scene_pcl.shape[1]))
    radius = 0.1
    ok = True
    while len(surface) == 0:
        near = scene_pcl[d2 <= radius ** 2]
        parts = [near[near[:, 4] == s] for s in ok_surface]                 # :125-131, in list order
        surface = np.concatenate(parts, axis=0) if parts else near[:0]
        surface = surface[surface[:, 2] > -3]                               # :133-134
        radius += 0.1                                                       # :136
        if radius > 5:                                                      # :138-140, also when found
            ok = False
            break
    out_anno = sample_anno
    if ok:
        acc = 0.0
        for v in surface[:, 2]:                                             # np.mean(axis=0): row by row
            acc += float(v)
        road_level = acc / len(surface)                                     # :144
        z_move = road_level - cz                                            # :146
        sample_pcl[:, 2] += z_move                                          # :147
        out_anno = make_annotation([cx, cy, road_level], anno_quat(sample_anno), sample_anno["length"],
                                   sample_anno["width"], sample_anno["height"], sample_anno["class"])
    return sample_pcl, out_anno, ok


def on_allowed_surface(sample_pcl, map_, map_move, transformation_matrix, ok_map_surface):
    """find_spot.py:234-248: every sample point that falls inside the map lies on an allowed cell
    (vacuously true when none does)."""
    hom = np.hstack((sample_pcl[:, :3], np.ones((len(sample_pcl), 1)))).T
    T = np.asarray(transformation_matrix, dtype=np.float64)
    g = blas_matvec4(T, hom[:, 0])[:, None] if hom.shape[1] == 1 else blas_matmul(T, hom)   # :235
    g = (g - map_move).astype(np.int64)                                            # :237-238 (np.int)
    g = g[:, g[0] < len(map_)]
    g = g[:, g[0] > -1]
    g = g[:, g[1] < len(map_[0])]
    g = g[:, g[1] > -1]
    cells = np.asarray(map_)[g[0], g[1]]
    return bool(np.isin(cells, ok_map_surface).all())


def placement_surfaces(sample_class, placement, placement_labels):
    """find_spot.py:218-223: map values and semantic labels an object of this class may stand on."""
    ok_map_surface = list(placement[int(sample_class)])
    ok_surface = []
    for m in ok_map_surface:
        ok_surface = ok_surface + list(placement_labels[m])
    return ok_map_surface, ok_surface


def find_possible_places(point_cloud, scene_annotation, sample_pcl, sample_anno_line, map_, map_move, original_pcl,
                         transformation_matrix, placement, placement_labels):
    """find_spot.py:192-273.  Returns (list of M x 5 candidate clouds, list of annotations, list of
    rotation numbers 1..360, not_on_road, object_collision).  The sample's points, its box centre
    and orientation are advanced by one degree per step and *keep* every height correction."""
    sample_pcl = np.array(sample_pcl, dtype=np.float64, copy=True)
    anno = read_label_line(sample_anno_line)
    ok_map_surface, ok_surface = placement_surfaces(anno["class"][0], placement, placement_labels)
    out_pcl, out_anno, out_rot = [], [], []
    not_on_road = object_collision = 0
    for rot in range(1, 361):
        sample_pcl, anno = rotate_bounding_box_2(sample_pcl, anno)                       # :230
        if not on_allowed_surface(sample_pcl, map_, map_move, transformation_matrix, ok_map_surface):
            not_on_road += 1
            continue
        sample_pcl, anno, near_road = correct_height(original_pcl, sample_pcl, anno, ok_surface)   # :251
        if not near_road:
            continue
        if check_bounding_box(point_cloud, scene_annotation, sample_pcl, anno, ok_surface):       # :256
            out_pcl.append(sample_pcl.copy())
            out_anno.append(anno)
            out_rot.append(rot)
        else:
            object_collision += 1
    return out_pcl, out_anno, out_rot, not_on_road, object_collision


# ==== object-detection flavour: object_detection/Real3DAug/tools/find_spot.py ========================
def read_label_line_od(line):
    """OD find_spot.py:179-224: a KITTI label_2 line (camera frame) -> annotation in the LiDAR frame."""
    it = line.split(" ")
    height, width, length = float(it[8]), float(it[9]), float(it[10])
    x, y, z = float(it[11]), float(it[12]), float(it[13])
    rot_y = float(it[14])
    c_height, c_width, c_length = height + 0.1, length + 0.1, width + 0.1            # :198-200
    cx, cy, cz = float(z) + 0.27, float(x) * -1, float(y) * -1 - 0.08                # :202-204
    a = float(rot_y) * -1
    m = np.array([[math.cos(a), -1 * math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], dtype=np.float64)
    q = matrix_to_quat(m)
    return make_annotation([cx, cy, cz], q, c_length, c_width, c_height, it[0])      # OD make_dictionary keeps the string


def rotate_bounding_box_od(bbox_pcl, annotation):
    """OD find_spot.py:72-105: as rotate_bounding_box_2, but the points are turned one by one with
    np.dot(z_rot_matrix, column) -- the BLAS matrix x vector kernel, which accumulates in another
    order than the matrix x matrix kernel (module docstring)."""
    rot = quat_to_matrix(quat_normalize(anno_quat(annotation)))
    q = matrix_to_quat(blas_matmul(rot, Z1))
    c = blas_matvec3(Z1, anno_center(annotation))
    x, y, z = bbox_pcl[:, 0].copy(), bbox_pcl[:, 1].copy(), bbox_pcl[:, 2].copy()
    for i in range(3):                                                                # :94-99
        bbox_pcl[:, i] = fma(Z1[i, 2], z, fma(Z1[i, 0], x, Z1[i, 1] * y))
    return bbox_pcl, make_annotation(c, q, annotation["length"], annotation["width"], annotation["height"],
                                     annotation["class"])


def on_road_od(sample_pcl, map_, map_move):
    """OD find_spot.py:261-275: some sample point falls inside the map and every one that does lies
    on a cell of value 1."""
    g0 = sample_pcl[:, 0] - map_move[0]
    g1 = sample_pcl[:, 1] - map_move[1]
    inside = ~((g0 < 0) | (g0 >= map_.shape[0]) | (g1 < 0) | (g1 >= map_.shape[1]))
    if not inside.any():
        return False
    cells = np.asarray(map_)[g0[inside].astype(np.int64), g1[inside].astype(np.int64)]
    return bool((cells == 1).all())


def check_bounding_box_od(scene_pcl, scene_anno, sample_pcl, sample_anno):
    """OD find_spot.py:108-135: scene points of label 1 inside the box collide; under a pedestrian
    only those at least 0.1 m above the box bottom (:123-124)."""
    inside = scene_pcl[box_mask(scene_pcl, sample_anno)]
    inside = inside[inside[:, 7] == 1]
    if sample_anno["class"] == "Pedestrian":
        inside = inside[inside[:, 2] >= sample_anno["center"]["z"] + 0.1]
    if len(inside):
        return False
    for a in scene_anno:
        if box_mask(sample_pcl, a).any():
            return False
    return True


def find_possible_places_od(point_cloud, scene_annotation, sample_pcl, sample_anno_line, map_, map_move, original_pcl,
                            road_label):
    """OD find_spot.py:227-304.  map_move = (min_x, min_y); road_label = config['labels']['Road']."""
    sample_pcl = np.array(sample_pcl, dtype=np.float64, copy=True)
    sample_pcl[:, 4] = 1                                                              # :249
    anno = read_label_line_od(sample_anno_line)
    out_pcl, out_anno, out_rot = [], [], []
    not_on_road = object_collision = 0
    for rot in range(1, 361):
        sample_pcl, anno = rotate_bounding_box_od(sample_pcl, anno)
        if not on_road_od(sample_pcl, map_, map_move):
            not_on_road += 1
            continue
        sample_pcl, anno, near_road = correct_height(original_pcl, sample_pcl, anno, [road_label])   # :155-158
        if not near_road:
            not_on_road += 1                                                          # :280
            continue
        if check_bounding_box_od(point_cloud, scene_annotation, sample_pcl, anno):
            out_pcl.append(sample_pcl.copy())
            out_anno.append(anno)
            out_rot.append(rot)
        else:
            object_collision += 1
    return out_pcl, out_anno, out_rot, not_on_road, object_collision
