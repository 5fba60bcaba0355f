"""
Per-sweep record keeping: the record_dict of reference pybgmm/gmm/gmm.py:45-118.

``sample_time`` keeps the reference's meaning: wall time of the sweep only (the
timer is restarted after the record is taken, igmm/crpmm.py:91-92).  ``log_marg``,
``components`` and ``nk`` come from device state.  The clustering metrics
(nmi / mi / vi / loss / bic) come from a device contingency-table kernel
and the components' sufficient statistics when ``self.record_metrics`` is True, and are
recorded as NaN otherwise.
"""
import logging
import time

import numpy as np  # noqa: F401

from ..utils import metrics as _metrics

logger = logging.getLogger(__name__)

RECORD_KEYS = ("sample_time", "log_marg", "components", "nmi", "mi", "nk", "loss", "bic", "vi",
               "alpha")


class GMM(object):
    record_metrics = True

    def __init__(self):
        pass

    def label_switch(self, idx, nplist):
        return np.array(nplist)[idx]

    def setup_record_dict(self):
        return dict((key, []) for key in RECORD_KEYS)

    def _clustering_metrics(self, true_assignments):
        """nmi, mi, vi (base 2) and the int-truncated inertia of the current labelling.  With the
        components on the GPU the label vector never leaves the device: a contingency-table
        kernel and the sufficient statistics provide everything; labels
        sort like ``np.unique`` on both sides, so the values equal the host formulas'."""
        ctx = getattr(self.components, "_ctx", None)
        if ctx is None:
            z = self.components.assignments
            return (_metrics.normalized_mutual_information(true_assignments, z),
                    _metrics.mutual_information(true_assignments, z),
                    _metrics.information_variation(true_assignments, z, base=2),
                    _metrics.cluster_loss_inertia(self.components.X, z))
        t = np.asarray(true_assignments)
        cache = getattr(self, "_true_idx_cache", None)
        if cache is None or cache[0] is not true_assignments:
            _metrics._check(t, t)
            uniq, idx = np.unique(t, return_inverse=True)
            cache = [true_assignments, idx.astype(np.int64), len(uniq), None]
            self._true_idx_cache = cache
        # the class indices go to the device once per (labelling, context)
        first = cache[3] is not ctx
        table = ctx.contingency(cache[1] if first else None, cache[2])
        cache[3] = ctx
        nmi, mi, vi = _metrics.table_metrics(table)
        loss = _metrics.loss_from_dispersion(ctx.cluster_dispersion())
        return nmi, mi, vi, loss

    def update_record_dict(self, record_dict, i_iter, true_assignments, start_time):
        record_dict["sample_time"].append(time.time() - start_time)
        record_dict["log_marg"].append(self.log_marg())
        K = self.components.K
        record_dict["components"].append(K)
        counts = self.components.counts[:K]
        if self.record_metrics and true_assignments is not None:
            nmi, mi, vi, loss = self._clustering_metrics(true_assignments)
        else:
            nmi = mi = loss = vi = float("nan")
        record_dict["nmi"].append(nmi)
        record_dict["mi"].append(mi)
        record_dict["nk"].append(str(counts))
        record_dict["loss"].append(loss)
        record_dict["bic"].append(loss)     # the reference records the same quantity twice (gmm.py:96-101)
        record_dict["vi"].append(vi)
        record_dict["alpha"].append(self.alpha)
        if i_iter % 20 == 0:
            info = "iteration: " + str(i_iter)
            for key in sorted(record_dict):
                info += ", " + key + ": " + str(record_dict[key][-1])
            info += "."
            logger.info(info)
        return record_dict
