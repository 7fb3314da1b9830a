"""CPU: the crafted parity cases of tests/tie_cases.py really contain what they are named after (depth
ties under the strict `<` of insertion.py:467, the row-0 truncation of :104-108), judged on the oracle's
intermediates; tests/test_gpu_ties.py then compares the HIP path with the oracle on them."""
import numpy as np

import tie_cases as T
from oracle import real3d_oracle as O


def _chain(case):
    xyzi, label, slots, need = case[:4]
    s5 = np.hstack((xyzi.astype(np.float64), (label & 0xFFFF).astype(np.float64)[:, None]))
    return O.augment_scene(s5, slots, need)


def test_coincident_points_tie_and_stay_hidden(synth):
    case = T.coincident_case(synth)
    n_patch = case[4]
    ties = T.premise_coincident(case)
    assert ties >= 30                                   # pixels where sample depth == scene depth, bit for bit
    merged, added, acc = _chain(case)
    assert acc == [0]
    # only the copies that share a pixel with a closer sample point are visible: a `<=` would add the tied ones
    assert len(added) == len(case[2][0][0]) - ties


def test_hole_mean_equal_to_sample_depth_stays_hidden(synth):
    case = T.hole_mean_case(synth)
    assert T.premise_hole_mean(case)
    merged, added, acc = _chain(case)
    assert acc == [0] and len(added) == 4               # the tie point is not among the visible ones
    assert len(merged) == len(case[0]) - 1 + 4


def test_row0_truncation_case_has_both_kinds(synth, monkeypatch):
    monkeypatch.setattr(O, "NUMROW", 448)
    monkeypatch.setattr(O, "NUMCOLUMN", 2880)
    in_row0, skipped = T.premise_row0(T.row0_edge_case(synth))
    assert in_row0 >= 100 and skipped >= 40


def test_odd_values_survive_the_oracle(synth):
    case = T.odd_values_case(synth)
    merged, added, acc = _chain(case)
    assert acc == [0] and np.isnan(added[:, 6]).any() and (added[:, 7] == 70001.0).all()
    vb, lb, cb = O.save_bytes_semantic(merged, added)
    lab = np.frombuffer(lb, dtype=np.uint32)
    assert (lab[-len(added):] == 70001).all() and lab[: len(merged) - len(added)].max() < 65536
