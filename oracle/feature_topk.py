"""CPU restatement of the reference's per-feature top-k tracker -- TEST INFRASTRUCTURE, never imported by the product.

Follows /root/reference/src/whisper_sae/analysis/feature_viz.py:94-158 (``TopKTracker.update``): walk batch row,
position, active feature in that order; an activation is ``value > 0`` (:129); a full list takes a new value only
when it is strictly greater than its smallest (:153); lists are read strongest first (:170).  Pinned by golden set
G13 (tests/golden/g13_feature_topk.npz, written by the reference's own TopKTracker).  Equal values: arrival order
decides, the earlier one first (the reference raises TypeError when equal values meet in a heap; G13 holds none).
"""

from __future__ import annotations

import numpy as np


class FeatureTopK:
    def __init__(self, num_features: int, k: int):
        self.num_features, self.k = num_features, k
        self.lists = [[] for _ in range(num_features)]  # (value, ordinal), strongest first
        self.total_activations = 0
        self.rows_seen = 0

    def _offer(self, f: int, v: float, o: int) -> None:
        lst = self.lists[f]
        if len(lst) == self.k and not v > lst[-1][0]:
            return
        j = 0
        while j < len(lst) and (lst[j][0] > v or (lst[j][0] == v and lst[j][1] < o)):
            j += 1
        lst.insert(j, (v, o))
        del lst[self.k:]

    def update_dense(self, a: np.ndarray) -> None:
        """a: [rows, H]; row r is ordinal rows_seen + r."""
        rows = a.shape[0]
        r_idx, f_idx = np.nonzero(a > 0)
        for r, f in zip(r_idx.tolist(), f_idx.tolist()):
            self._offer(f, float(a[r, f]), self.rows_seen + r)
        self.total_activations += len(r_idx)
        self.rows_seen += rows

    def update_compact(self, vals: np.ndarray, idx: np.ndarray) -> None:
        """vals/idx: [rows, k_code]; the reference sees the same entries in ascending feature order per row."""
        rows = vals.shape[0]
        for r in range(rows):
            order = np.argsort(idx[r], kind="stable")
            for j in order.tolist():
                if vals[r, j] > 0:
                    self._offer(int(idx[r, j]), float(vals[r, j]), self.rows_seen + r)
                    self.total_activations += 1
        self.rows_seen += rows

    def arrays(self):
        v = np.zeros((self.num_features, self.k), np.float32)
        o = np.zeros((self.num_features, self.k), np.int64)
        c = np.zeros(self.num_features, np.int32)
        for f, lst in enumerate(self.lists):
            c[f] = len(lst)
            for j, (val, ordinal) in enumerate(lst):
                v[f, j], o[f, j] = val, ordinal
        return v, o, c
