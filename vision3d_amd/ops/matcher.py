"""IoU-band label assignment (interface of `vision3d/ops/matcher.py:6-130`).

Given an (M gt) x (N anchors) quality matrix, every anchor gets the index of its best gt and a label
from the band its best quality falls in: thresholds [t0, t1, ...] split [-inf, inf) into
len(labels) half-open bands [low, high).  Kept in PyTorch: it is a column max plus a table lookup.
"""
from typing import List

import torch


class Matcher(object):

    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        assert thresholds[0] > 0
        assert all(a <= b for a, b in zip(thresholds[:-1], thresholds[1:]))
        assert len(labels) == len(thresholds) + 1
        assert all(l in (-1, 0, 1) for l in labels)
        self.thresholds = [-float("inf")] + list(thresholds) + [float("inf")]
        self.labels = list(labels)
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        q = match_quality_matrix
        assert q.dim() == 2
        n = q.size(1)
        if q.numel() == 0:
            # no gt: every anchor falls in the lowest band (matcher.py:69-79)
            return (q.new_zeros((n,), dtype=torch.int64),
                    q.new_full((n,), self.labels[0], dtype=torch.int8))
        assert torch.all(q >= 0)
        best, matches = q.max(dim=0)
        edges = torch.tensor(self.thresholds[1:-1], dtype=best.dtype, device=best.device)
        band = torch.bucketize(best, edges, right=True)  # edges[b-1] <= best < edges[b]
        table = torch.tensor(self.labels, dtype=torch.int8, device=best.device)
        match_labels = table[band]
        if self.allow_low_quality_matches:
            self.set_low_quality_matches_(match_labels, q)
        return matches, match_labels

    def set_low_quality_matches_(self, match_labels, match_quality_matrix):
        """Every anchor that attains some gt's best quality (ties included) becomes positive
        (Faster R-CNN 3.1.2 case (i); matcher.py:98-130)."""
        top_per_gt = match_quality_matrix.max(dim=1, keepdim=True).values
        hit = (match_quality_matrix == top_per_gt).any(dim=0)
        match_labels[hit] = 1


def subsample_labels(labels, num_samples, positive_fraction, bg_label):
    """Random pos/neg index subsample (matcher.py:133-174); unused by the detector, kept for API parity."""
    pos = torch.nonzero((labels != -1) & (labels != bg_label)).squeeze(1)
    neg = torch.nonzero(labels == bg_label).squeeze(1)
    n_pos = min(pos.numel(), int(num_samples * positive_fraction))
    n_neg = min(neg.numel(), num_samples - n_pos)
    pos = pos[torch.randperm(pos.numel(), device=pos.device)[:n_pos]]
    neg = neg[torch.randperm(neg.numel(), device=neg.device)[:n_neg]]
    return pos, neg
