"""
Per-sweep clustering metrics of the record dict, computed from a K_true x K
contingency table instead of the reference's O(K_true*K*N) scans.

Values follow reference pybgmm/infopy/infopy.py:19-119 and
pybgmm/utils/utils.py:31-88 including their quirks:
  * NMI normaliser ``max(sqrt(H_true*H_pred), 1e-10)`` with NATURAL-log entropies
    even when ``base`` is given (infopy.py:95-96);
  * ``entropy`` over ``np.bincount`` probabilities, zero bins skipped (infopy.py:19-29);
  * the inertia "loss" is the sum of per-cluster ``sqrt(sum (x-mean)^2)`` values
    TRUNCATED to integers, because the reference stores them in an int array
    (utils.py:39-48).
(SURVEY.md 8f rank 2.  ``table_metrics`` takes the K_true x K table that ``bgmm_contingency`` builds on
the device and the per-cluster dispersions of ``bgmm_cluster_dispersion``; the label-vector functions
are the same formulas for callers that hold labels on the host.)
"""
import math

import numpy as np

E = 2.718281828459045


def _check(labels_true, labels_pred):
    a, b = np.asarray(labels_true), np.asarray(labels_pred)
    if a.ndim != 1:
        raise ValueError("labels_true must be 1D: shape is %r" % (a.shape,))
    if b.ndim != 1:
        raise ValueError("labels_pred must be 1D: shape is %r" % (b.shape,))
    if a.shape != b.shape:
        raise ValueError("labels_true and labels_pred must have same size, got %d and %d"
                         % (a.shape[0], b.shape[0]))
    return a, b


def entropy(x, base=E):
    if len(x) == 0:
        return 1.0
    p = np.bincount(x) / float(len(x))
    total = 0
    for p_i in p:
        if p_i == 0:
            continue
        total -= p_i * math.log(p_i, base)
    return total


def mutual_information(labels_true, labels_pred, normalized=False, base=E):
    a, b = _check(labels_true, labels_pred)
    n = len(a)
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    table = np.zeros((len(ua), len(ub)), dtype=np.int64)
    np.add.at(table, (ia, ib), 1)
    ca, cb = table.sum(axis=1), table.sum(axis=0)
    mi = 0.0
    for r in range(len(ua)):
        px = ca[r] / n
        for c in np.nonzero(table[r])[0]:
            pxy = table[r, c] / n
            py = cb[c] / n
            mi += pxy * math.log((pxy / (px * py)), base)
    if normalized:
        mi = mi / max(np.sqrt(entropy(a) * entropy(b)), 1e-10)
    return mi


def normalized_mutual_information(labels_true, labels_pred, base=E):
    return mutual_information(labels_true, labels_pred, normalized=True, base=base)


def information_variation(labels_true, labels_pred, base=E):
    a, b = _check(labels_true, labels_pred)
    return entropy(a, base=base) + entropy(b, base=base) - (2 * mutual_information(a, b, base=base))


def cluster_loss_inertia(x, assignments):
    assignments = np.asarray(assignments)
    labels = np.unique(assignments)
    total = np.zeros((), dtype=labels.dtype)
    for lab in labels:
        x_k = x[np.where(assignments == lab)[0], :]
        mean = np.sum(x_k, axis=0) / float(x_k.shape[0]) if x_k.shape[1] <= 2 else np.mean(x_k, axis=0)
        dist = np.sqrt(np.sum(np.square(x_k - mean)))
        total = total + np.asarray(dist).astype(labels.dtype)   # int truncation, as the reference
    return total


# ---- the same quantities from a contingency table (device path) ------- #
def _entropy_from_counts(counts, n, base=E):
    total = 0
    for c in counts:
        if c == 0:
            continue
        p_i = c / float(n)
        total -= p_i * math.log(p_i, base)
    return total


def table_metrics(table):
    """(nmi, mi, vi_base2) from a K_true x K contingency table whose rows / columns are ordered by
    increasing label value -- the values ``normalized_mutual_information``,
    ``mutual_information`` and ``information_variation(base=2)`` return for the label vectors."""
    table = np.asarray(table, dtype=np.int64)
    n = int(table.sum())
    ca, cb = table.sum(axis=1), table.sum(axis=0)

    def mi(base):
        acc = 0.0
        for r in range(table.shape[0]):
            if ca[r] == 0:
                continue
            px = ca[r] / n
            for c in np.nonzero(table[r])[0]:
                pxy = table[r, c] / n
                py = cb[c] / n
                acc += pxy * math.log((pxy / (px * py)), base)
        return acc

    mi_e = mi(E)
    nmi = mi_e / max(np.sqrt(_entropy_from_counts(ca, n) * _entropy_from_counts(cb, n)), 1e-10)
    vi = _entropy_from_counts(ca, n, 2) + _entropy_from_counts(cb, n, 2) - 2 * mi(2)
    return nmi, mi_e, vi


def loss_from_dispersion(dispersion, dtype=np.int64):
    """``cluster_loss_inertia``: per-cluster sqrt of the summed squared distances to the mean,
    truncated to integers (the reference stores them in an int array) and summed."""
    total = np.zeros((), dtype=dtype)
    for v in np.sqrt(np.asarray(dispersion, dtype=np.float64)):
        total = total + np.asarray(v).astype(dtype)
    return total
