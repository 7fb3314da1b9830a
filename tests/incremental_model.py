"""NumPy model of the INCREMENTAL batched pipeline the HIP kernels implement (test infrastructure).

The reference re-projects the whole merged cloud for every insert (insertion.py:371-377).  The
device pipeline does not: while a scene's elevation bounds stay bitwise the same, a surviving
point keeps its pixel id, so one insert step only has to

* evaluate the sample against the scene's range image on the sample's candidate pixels,
* overwrite the scene range image at the visible pixels (every scene point there is culled and
  the visible sample points, which all lie in those pixels, take their place),
* remember per pixel the last step at which it was visible (``stamp``) -- a point is dead at the
  end iff its pixel was visible at a step later than the point's birth,
* append the visible points to the tail log.

The cloud is compacted (dead points dropped) once at the end, or earlier when the bounds may
have moved ("rebase": the pixel holding the recorded max- or min-elevation point was culled, or
a visible point lies outside the old bounds); after a rebase everything is re-projected like step 0.

This file states that algorithm in plain NumPy so that ``tests/test_incremental_model.py`` can
prove it equal to the oracle's literal K-insert chain on the CPU, before any kernel runs; the
HIP kernels in ``csrc/r3d_batch.hip`` follow it function by function.
"""
import numpy as np

from oracle import real3d_oracle as O

SENT = np.inf   # empty pixel marker of the model (device: all-ones u64 key)


class IncrementalScene:
    def __init__(self, xyzi, label, num_row=O.NUMROW, num_column=O.NUMCOLUMN):
        self.R, self.C = num_row, num_column
        n = len(xyzi)
        # cloud = head (float32-exact points) followed by tail (float64 inserted points)
        self.xyz = xyzi[:, :3].astype(np.float64)
        self.inten = xyzi[:, 3].astype(np.float64)
        self.label = label.astype(np.float64)
        self.birth = np.zeros(n, dtype=np.int64)
        self.log = np.zeros((0, 9))          # all_visible_parts (insertion.py:534-545), append only
        self.step_no = 0
        self.rebases = 0
        self._project_all()

    # -- step 0 and rebase: the full pass (insertion.py:373-375 on the current cloud) ------------
    def _project_all(self):
        pc = np.full((len(self.xyz), 9), -1.0)
        pc[:, :3] = self.xyz
        pc, self.max_el, self.min_el = O.fill_spherical(pc)
        train, lab, pc = O.geometrical_front_view(pc, self.R, self.C, self.max_el, self.min_el)
        self.pix = pc[:, 8].astype(np.int64)
        self.grid = np.where(lab == 1, train, SENT)          # raw min depth, SENT where empty
        self.stamp = np.zeros((self.R, self.C), dtype=np.int64)
        # pixel of one max-elevation point and of one min-elevation point: while those two pixels
        # stay uncovered their points survive and the bounds cannot shrink
        self.extreme_pix = (int(self.pix[np.argmax(pc[:, 5])]), int(self.pix[np.argmin(pc[:, 5])]))

    def _alive(self):
        st = self.stamp.reshape(-1)[(self.pix // O.NUMCOLUMN) * self.C + self.pix % O.NUMCOLUMN]
        return st <= self.birth

    def _compact(self):
        keep = self._alive()
        self.xyz, self.inten, self.label = self.xyz[keep], self.inten[keep], self.label[keep]
        self.birth, self.pix = self.birth[keep], self.pix[keep]

    # -- one insert step ------------------------------------------------------------------------
    def step(self, sample5, need):
        self.step_no += 1
        R, C = self.R, self.C
        smp = O.add_space_for_spherical(np.asarray(sample5, dtype=np.float64))
        smp, _, _ = O.fill_spherical(smp)
        s_train, s_lab, smp = O.geometrical_front_view(smp, R, C, self.max_el, self.min_el, sample=True)
        s_train2, _ = O.smooth_out(s_train, s_lab)
        sc_train = np.where(self.grid == SENT, O.EMPTY_DEPTH, self.grid)
        sc_lab = np.where(self.grid == SENT, -1.0, 1.0)
        sc_train2, _ = O.smooth_out(sc_train, sc_lab)
        vis = s_train2 < sc_train2
        spix = smp[:, 8].astype(np.int64)
        valid = spix >= 0
        hit = np.zeros(len(smp), dtype=bool)
        hit[valid] = vis.reshape(-1)[(spix[valid] // O.NUMCOLUMN) * C + spix[valid] % O.NUMCOLUMN]
        order = np.argsort(spix[hit], kind="stable")
        visible = smp[hit][order]
        if len(visible) == 0 or len(visible) < need:
            return 0
        # commit ---------------------------------------------------------------------------
        rr, cc = np.nonzero(vis)
        culled = (rr * self.C + cc)[self.grid[rr, cc] != SENT]
        rebase = bool(np.isin(np.array(self.extreme_pix), culled).any())
        rebase |= bool(np.any(visible[:, 5] < self.min_el) or np.any(visible[:, 5] > self.max_el))
        self.grid[rr, cc] = np.where(s_lab[rr, cc] == 1, s_train[rr, cc], SENT)
        self.stamp[rr, cc] = self.step_no
        self.xyz = np.vstack([self.xyz, visible[:, :3]])
        self.inten = np.concatenate([self.inten, visible[:, 6]])
        self.label = np.concatenate([self.label, visible[:, 7]])
        self.birth = np.concatenate([self.birth, np.full(len(visible), self.step_no)])
        self.pix = np.concatenate([self.pix, visible[:, 8].astype(np.int64)])
        self.log = np.vstack([self.log, visible])
        if rebase:
            self.rebases += 1
            self._compact()
            self._project_all()
        return len(visible)

    def finalize(self):
        """(merged x y z intensity label [n,5] float64, merged pix, all_visible_parts [m,9])."""
        self._compact()
        merged = np.column_stack([self.xyz, self.inten, self.label])
        return merged, self.pix, self.log
