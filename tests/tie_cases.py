"""Crafted inputs for the parity soft spots: depth ties under the strict `<`
of insertion.py:467, non-finite / negative intensities, labels above 16 bits, and the row-0 truncation
edge (insertion.py:104, int() rounds toward zero) on a 448 x 2880 grid.

Every case is (xyzi float32 [n,4], label uint32 [n], slots, need) like tests/test_gpu_batch.py's
cases; `premise_*` functions state, on the oracle's intermediates, that the case really contains the
situation it is named after (checked in the CPU suite), the GPU suite then compares bytes with the oracle.
"""
import numpy as np

from oracle import real3d_oracle as O


def _scene9(xyzi, label):
    s5 = np.hstack((xyzi.astype(np.float64), (label & 0xFFFF).astype(np.float64)[:, None]))
    return O.add_space_for_spherical(s5)


def _views(xyzi, label, sample5, rows=None, cols=None):
    """(scene_train_closed, sample_train_closed, scene_raw, sample_raw, scene9, sample9) of one candidate."""
    rows = O.NUMROW if rows is None else rows
    cols = O.NUMCOLUMN if cols is None else cols
    scene = _scene9(xyzi, label)
    scene, max_el, min_el = O.fill_spherical(scene)
    raw, lab, scene = O.geometrical_front_view(scene, rows, cols, max_el, min_el)
    closed, _ = O.smooth_out(raw, lab)
    smp = O.add_space_for_spherical(np.asarray(sample5, dtype=np.float64))
    smp, _, _ = O.fill_spherical(smp)
    sraw, slab, smp = O.geometrical_front_view(smp, rows, cols, max_el, min_el, sample=True)
    sclosed, _ = O.smooth_out(sraw, slab)
    return closed, sclosed, raw, sraw, scene, smp


# ---- 1. a sample point coincident with a scene point ------------------------------------------------
def coincident_case(synth, seed=7):
    """The sample is a patch of the scene's own points (same float32 coordinates, so the same pixel and the
    same depth: `sample_train < scene_train` is false there and the copies stay hidden) plus a slab of
    points 20 % closer in front of one half of the patch (visible)."""
    xyzi, label = synth.make_scene(seed, 32, 600)
    xyz = xyzi[:, :3].astype(np.float64)
    az = np.arctan2(xyz[:, 1], xyz[:, 0])
    el = np.arccos(xyz[:, 2] / np.sqrt((xyz * xyz).sum(1)))
    patch = np.nonzero((np.abs(az - 0.4) < 0.06) & (el > 1.62) & (el < 1.80))[0]
    copies = np.column_stack([xyz[patch], np.full(len(patch), 0.5), np.full(len(patch), 30.0)])
    front = copies[: len(copies) // 2].copy()
    front[:, :3] *= 0.8
    sample = np.vstack([copies, front])
    return xyzi, label, [[sample]], [5], len(patch)


def premise_coincident(case):
    xyzi, label, slots, _, n_patch = case
    closed, sclosed, raw, sraw, scene, smp = _views(xyzi, label, slots[0][0])
    pix = smp[:n_patch, 8].astype(np.int64)
    ties = (sraw.ravel()[pix] == raw.ravel()[pix]) & (raw.ravel()[pix] < 500.0)
    return int(ties.sum())


# ---- 2. a hole of the scene whose mean equals the sample's depth there ---------------------------------
def _equal_norm_neighbours(cols=1440, want=4):
    """Integer vectors (a, b) != (a2, b2) with a*a + b*b == a2*a2 + b2*b2 whose azimuths fall into neighbouring
    columns: scaled by a power of two they are exact float32 coordinates with exactly equal ranges."""
    d_az = 2 * np.pi / cols
    a = np.arange(1200, 2000)[:, None]
    b = np.arange(1, 900)[None, :]
    n2 = (a * a + b * b).ravel()
    aa, bb = np.broadcast_to(a, (a.shape[0], b.shape[1])).ravel(), np.broadcast_to(b, (a.shape[0], b.shape[1])).ravel()
    order = np.argsort(n2, kind="stable")
    n2, aa, bb = n2[order], aa[order], bb[order]
    out = []
    same = np.nonzero(n2[1:] == n2[:-1])[0]
    for i in same:
        for j in range(i + 1, min(i + 6, len(n2))):
            if n2[j] != n2[i]:
                break
            c1 = int((np.arctan2(bb[i], -aa[i]) + np.pi) % (2 * np.pi) / d_az)
            c2 = int((np.arctan2(bb[j], -aa[j]) + np.pi) % (2 * np.pi) / d_az)
            if abs(c1 - c2) == 1:
                out.append(((int(aa[i]), int(bb[i])), (int(aa[j]), int(bb[j]))))
                if len(out) >= want:
                    return out
    return out


def hole_mean_case(synth, seed=9):
    """Two columns of the scene are empty between two occupied ones (the 3-wide closing fills both); the hole next to
    the left one has ONE occupied neighbour, a single point, so its mean is that point's depth; the sample puts a
    point of exactly that depth (an integer vector of the same norm) into the hole: a tie under the strict `<`, the
    sample's pixel stays hidden.  Four sample points in front of the left neighbour are visible."""
    pairs = _equal_norm_neighbours()
    assert pairs, "no equal-norm lattice pair found"
    (a, b), (a2, b2) = pairs[0]
    scale = 2.0 ** -7                                          # ~12 m
    # background: a small scan that fixes the elevation bounds well away from the crafted rows
    xyzi, label = synth.make_scene(seed, 16, 300)
    keep = np.hypot(xyzi[:, 0], xyzi[:, 1]) > 0               # all
    xyzi, label = xyzi[keep], label[keep]
    # drop background points near the crafted azimuth so that the crafted pixels hold only crafted points
    az_c = np.arctan2(b, -a)
    az = np.arctan2(xyzi[:, 1].astype(np.float64), xyzi[:, 0].astype(np.float64))
    far = np.abs(((az - az_c + np.pi) % (2 * np.pi)) - np.pi) > 0.08
    xyzi, label = xyzi[far], label[far]
    z = np.float32(-100.0 * scale)                              # slightly below the horizon
    left = np.array([-a * scale, b * scale, z, 0.25], dtype=np.float32)
    # a point two columns further right closes the gap: mirror of `left` about the hole (any depth)
    d_az = 2 * np.pi / 1440
    c_left = int((np.arctan2(b, -a) + np.pi) % (2 * np.pi) / d_az)
    c_tie = int((np.arctan2(b2, -a2) + np.pi) % (2 * np.pi) / d_az)
    az_r = (c_tie + 2 * (c_tie - c_left) + 0.5) * d_az - np.pi   # centre of the column two past the hole: a 2-wide gap
    rr = float(np.hypot(a, b)) * scale * 1.07
    right = np.array([rr * np.cos(az_r), rr * np.sin(az_r), z * 1.07, 0.75], dtype=np.float32)
    xyzi = np.vstack([xyzi, left[None], right[None]]).astype(np.float32)
    label = np.concatenate([label, np.array([50, 50], dtype=np.uint32)])
    tie = np.array([-a2 * scale, b2 * scale, float(z), 0.5, 30.0])
    toward_left = np.array([-a * scale, b * scale, float(z), 0.5, 30.0])   # in front of the left neighbour: visible
    sample = np.vstack([tie] + [toward_left * np.array([f, f, f, 1.0, 1.0]) for f in (0.9, 0.8, 0.7, 0.6)])
    return xyzi, label, [[sample]], [1]


def premise_hole_mean(case):
    xyzi, label, slots, _ = case
    closed, sclosed, raw, sraw, scene, smp = _views(xyzi, label, slots[0][0])
    p = int(smp[0, 8])
    hole = raw.ravel()[p] == 500.0 and closed.ravel()[p] < 500.0
    return bool(hole and closed.ravel()[p] == smp[0, 3])


# ---- 3. intensities that are NaN / negative / infinite, labels above 16 bits -------------------------------
def odd_values_case(synth, seed=11):
    xyzi, label = synth.make_scene(seed, 24, 500)
    xyzi = xyzi.copy()
    rng = np.random.default_rng(seed)
    k = rng.choice(len(xyzi), 300, replace=False)
    xyzi[k[:100], 3] = np.nan
    xyzi[k[100:200], 3] = -3.5
    xyzi[k[200:], 3] = np.inf
    label = label.copy()
    label[k] |= np.uint32(0x00AB0000)                           # instance bits above the semantic label
    smp = synth.make_insert(seed, "pedestrian", rng_range=(5.0, 9.0))
    smp[::3, 3] = np.nan
    smp[1::3, 3] = -1.25
    smp[:, 4] = 70001.0                                         # a 17-bit class id
    return xyzi, label, [[smp]], [10]


# ---- 4. the row-0 truncation edge on 448 x 2880 ---------------------------------------------------------
def row0_edge_case(synth, seed=13, rows=448, cols=2880):
    """Sample points whose elevation lies a fraction of a row ABOVE the scene's highest beam: rows in (-1, 0)
    are truncated into row 0 (insertion.py:104-108), rows <= -1 are skipped."""
    xyzi, label = synth.make_scene(seed, 64, 700)
    xyz = xyzi[:, :3].astype(np.float64)
    r = np.sqrt((xyz * xyz).sum(1))
    el = np.arccos(xyz[:, 2] / r)
    min_el, max_el = el.min(), el.max()
    d_el = (max_el - min_el) / rows
    rng = np.random.default_rng(seed)
    n = 240
    frac = np.concatenate([rng.uniform(-0.98, -0.02, n // 2), rng.uniform(-2.5, -1.02, n // 4), rng.uniform(0.05, 3.0, n // 4)])
    e = min_el + 0.00001 + frac * d_el
    az = 0.7 + rng.uniform(-0.01, 0.01, len(e))
    rr = 9.0 + rng.uniform(-0.05, 0.05, len(e))
    pts = np.column_stack([rr * np.sin(e) * np.cos(az), rr * np.sin(e) * np.sin(az), rr * np.cos(e)])
    smp = np.column_stack([pts, rng.random(len(e)), np.full(len(e), 30.0)])
    return xyzi, label, [[smp]], [5]


def premise_row0(case, rows=448, cols=2880):
    xyzi, label, slots, _ = case
    closed, sclosed, raw, sraw, scene, smp = _views(xyzi, label, slots[0][0], rows, cols)
    placed = smp[:, 8] >= 0
    row = (smp[:, 8] // O.NUMCOLUMN)[placed]
    return int((row == 0).sum()), int((~placed).sum())
