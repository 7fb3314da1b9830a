#!/usr/bin/env python3
"""Randomised parity campaign: batches of random scenes, grids, insert chains (overlapping placements, far points,
samples outside the field of view, thresholds around the visible count, several candidates per slot, long chains)
through the HIP path and through the oracle; every scene's velodyne / label / check bytes and accept list must be
equal.  Test infrastructure (imports oracle/), beside the fixed cases of tests/test_gpu_batch.py.

    python tools/fuzz_parity.py [batches] [first_seed] [workers]

The oracle runs in worker processes started BEFORE this process touches the GPU.  Exit code 1 on any mismatch; the
failing (seed, scene) pairs are printed so that a case can be replayed with `python tools/fuzz_parity.py 1 <seed>`.
"""
import importlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GRIDS = [(112, 1440)] * 5 + [(64, 1024), (16, 352), (128, 2048), (32, 4000), (448, 2880)]
KINDS = ["pedestrian", "cyclist", "car"]


def make_case(synth, rng, rows, cols):
    """One scene and its chain: (xyzi, label, slots, need)."""
    beams = int(rng.choice([8, 16, 32, 48, 64]))
    naz = int(rng.integers(60, 1100))
    if rng.random() < 0.08:
        beams, naz = 64, 1875                                 # a whole KITTI-sized frame
    xyzi, label = synth.make_scene(int(rng.integers(1 << 30)), beams, naz, shuffle=bool(rng.integers(2)))
    if rng.random() < 0.15:                                   # a ragged cut: not a multiple of anything
        keep = int(rng.integers(1, len(xyzi)))
        xyzi, label = np.ascontiguousarray(xyzi[:keep]), np.ascontiguousarray(label[:keep])
    if rng.random() < 0.3:                                    # some returns beyond 500 m (visible to any insert)
        idx = rng.choice(len(xyzi), size=min(len(xyzi), int(rng.integers(1, 40))), replace=False)
        xyzi[idx, :3] *= np.float32(rng.uniform(15.0, 40.0))
    if rng.random() < 0.2:                                    # duplicated points: depth ties inside the scene
        idx = rng.choice(len(xyzi), size=min(len(xyzi), 50), replace=False)
        xyzi = np.ascontiguousarray(np.vstack((xyzi, xyzi[idx])))
        label = np.ascontiguousarray(np.concatenate((label, label[idx])))
    if rng.random() < 0.2:
        label = (label | (rng.integers(0, 1 << 16, size=len(label)).astype(np.uint32) << 16)).astype(np.uint32)   # instance ids
    nk = int(rng.choice([1, 2, 3, 5, 5, 8, 12, 20]))
    clustered = rng.random() < 0.5                            # placements around one azimuth: every slot conflicts
    az0 = rng.uniform(-np.pi, np.pi)
    slots, need = [], []
    for k in range(nk):
        n_cand = 1 if rng.random() < 0.8 else int(rng.integers(2, 4))
        cands = []
        for _ in range(n_cand):
            kind = KINDS[int(rng.integers(3))]
            pts = int(rng.choice([3, 30, 200, 600, 1500, 3000])) if rng.random() < 0.5 else None
            az = az0 + rng.normal(0.0, 0.08) if clustered else rng.uniform(-np.pi, np.pi)
            az = (az + np.pi) % (2 * np.pi) - np.pi
            smp = synth.make_insert(int(rng.integers(1 << 30)), kind, points=pts, centre_range=float(rng.uniform(2.2, 45.0)),
                                    centre_az=float(az))
            r = rng.random()
            if r < 0.05:
                smp[:, 2] += 30.0                             # above the field of view
            elif r < 0.10:
                smp[:, 2] -= 1.2                              # partly below the lowest beam
            elif r < 0.13:
                smp[:, :3] *= 200.0                           # far away: a handful of pixels
            cands.append(smp)
        slots.append(cands)
        need.append(int(rng.choice([0, 1, 5, 20, 20, 20, 60, 400])))
    return xyzi, label, slots, need


def oracle_case(args):
    rows, cols, xyzi, label, slots, need = args
    O = importlib.import_module("oracle.real3d_oracle")
    s5 = np.hstack((xyzi.astype(np.float64), (label & 0xFFFF).astype(np.float64)[:, None]))
    O.NUMROW, O.NUMCOLUMN = rows, cols                        # the reference's two globals (insertion.py:22-23): the pixel ids follow them
    merged, allvis, acc = O.augment_scene(s5, slots, need)
    vb, lb, cb = O.save_bytes_semantic(merged, allvis)
    return vb, lb, cb, acc


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    pool = mp.get_context("spawn").Pool(workers)              # before any GPU call in this process
    pkg = importlib.import_module("pcl-augmentation_amd")
    synth = pkg.synth
    bad, scenes_done, both_raised, t0 = [], 0, 0, time.time()
    edge_risk, points = [0, 0], [0, 0]      # points within 1e-12 of a bin edge the HIP path met (scene, sample) | points given to it
    pending = []
    for bi in range(n_batches):
        seed = seed0 + bi
        rng = np.random.default_rng(seed)
        rows, cols = GRIDS[int(rng.integers(len(GRIDS)))]
        B = int(rng.choice([1, 2, 3, 6, 9]))
        cases = [make_case(synth, rng, rows, cols) for _ in range(B)]
        want = pool.map_async(oracle_case, [(rows, cols) + c for c in cases])
        # the default launch | every speculative evaluation verified | no speculation | kill masks without hits | every scene in
        # virtual order (the counting sort at begin, the way back to slab order at finish)
        debug = int(rng.choice([0, 0, 0, 64, 2, 128, 1024, 1024]))
        try:
            res, acc = pkg.augment_batch([(c[0], c[1]) for c in cases], [c[2] for c in cases], [c[3] for c in cases],
                                         rows=rows, cols=cols, debug=debug)
            err = None
            risk = pkg.SceneBatch.edge_risk_total                     # (a fresh batch per call: the call's own count)
            edge_risk[0] += risk[0]
            edge_risk[1] += risk[1]
            points[0] += sum(len(c[0]) for c in cases)
            points[1] += sum(len(x) for c in cases for slot in c[2] for x in slot)
        except Exception as e:                                # a status the oracle must explain (e.g. an assert of the reference)
            res, acc, err = None, None, e
        pending.append((seed, rows, cols, debug, cases, res, acc, err, want))
        while pending and (len(pending) > 3 or bi == n_batches - 1):
            seed_, rows_, cols_, debug_, cases_, res_, acc_, err_, want_ = pending.pop(0)
            try:
                oracle = want_.get()
            except Exception as oe:
                if err_ is None:
                    bad.append((seed_, -1, f"oracle raised {oe!r}, the HIP path did not"))
                else:
                    both_raised += 1                          # e.g. the reference's assert on an empty range image
                    print(f"seed {seed_}: both raised -- oracle {oe!r:.120}; HIP path {err_!r:.160}", flush=True)
                continue
            if err_ is not None:
                bad.append((seed_, -1, f"HIP path raised {err_!r}"))
                continue
            for i, (o, r, a) in enumerate(zip(oracle, res_, acc_)):
                vb, lb, cb, oacc = o
                ok = list(a) == list(oacc) and r[0].tobytes() == vb and r[1].tobytes() == lb and r[2].tobytes() == cb
                scenes_done += 1
                if not ok:
                    bad.append((seed_, i, f"grid {rows_}x{cols_} debug {debug_} accepted {list(a)} oracle {list(oacc)} "
                                          f"n_out {len(r[0])} oracle {len(vb) // 16}"))
            print(f"seed {seed_}: {rows_}x{cols_}, {len(cases_)} scenes, debug {debug_}; {scenes_done} scenes compared, "
                  f"{len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
    pool.close()
    for b in bad:
        print("MISMATCH", b)
    print(f"{scenes_done} scenes in {n_batches} batches: {len(bad)} mismatches; {both_raised} batches in which both sides raised")
    print(f"bin-edge risk: {edge_risk[0]} of {points[0]} scene points and {edge_risk[1]} of {points[1]} candidate points were decided by the "
          f"reference formula with the fractional row / column position within 1e-12 of an integer (all of them compared above)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
