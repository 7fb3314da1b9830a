"""GPU parity on the crafted soft spots (tests/tie_cases.py): depth ties under the strict `<`
(insertion.py:467) -- a sample point coincident with a scene point, a hole mean equal to the sample's
depth --, NaN / negative / infinite intensities, labels above 16 bits, the row-0 truncation edge on a
448 x 2880 grid.  Bytes of velodyne / labels / check must equal the oracle's."""
import numpy as np
import pytest

import tie_cases as T
from oracle import real3d_oracle as O

pytestmark = pytest.mark.gpu


def _oracle(case):
    xyzi, label, slots, need = case[:4]
    s5 = np.hstack((xyzi.astype(np.float64), (label & 0xFFFF).astype(np.float64)[:, None]))
    merged, allvis, acc = O.augment_scene(s5, slots, need)
    return O.save_bytes_semantic(merged, allvis), acc


def _run(pkg, cases, **kw):
    res, acc = pkg.augment_batch([(c[0], c[1]) for c in cases], [c[2] for c in cases], [c[3] for c in cases], **kw)
    for c, r, a in zip(cases, res, acc):
        (vb, lb, cb), oacc = _oracle(c)
        assert a == oacc
        assert r[0].tobytes() == vb and r[1].tobytes() == lb and r[2].tobytes() == cb


def test_depth_ties_and_odd_values(pkg, synth):
    cases = [T.coincident_case(synth), T.hole_mean_case(synth), T.odd_values_case(synth),
             T.coincident_case(synth, seed=21), T.odd_values_case(synth, seed=23)]
    _run(pkg, cases)
    # and one by one through the single-slot entry point (r3d_batch_insert)
    for c in cases[:3]:
        _run(pkg, [c])


def test_row0_truncation_on_448x2880(pkg, synth, monkeypatch):
    monkeypatch.setattr(O, "NUMROW", 448)
    monkeypatch.setattr(O, "NUMCOLUMN", 2880)
    _run(pkg, [T.row0_edge_case(synth), T.row0_edge_case(synth, seed=31)], rows=448, cols=2880)
