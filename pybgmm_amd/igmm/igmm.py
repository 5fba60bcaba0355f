"""
Base class of the infinite Gaussian mixture samplers -- the interface of reference
pybgmm/igmm/igmm.py:37-227 (``IGMM``), with the components on the GPU.
"""
import math

import numpy as np
from scipy import stats
from scipy.special import gammaln

from .. import _lib
from ..gaussian.gaussian_components import GaussianComponents, GaussianComponentsDiag, resume_in_larger_context
from ..gaussian.gaussian_components_fixedvar import GaussianComponentsFixedVar
from ..gmm import GMM
from ..utils import rng as _rng


def compact_labels(z):
    """Close gaps in a label vector keeping the order of the labels -- the net effect of
    the reference's shift-down loop (igmm.py:89-94)."""
    return np.unique(z, return_inverse=True)[1].astype(np.int64)


class IGMM(GMM):
    """
    An infinite Gaussian mixture model.

    X : N x D data.  kernel_prior : ``NIW``.  alpha : CRP concentration.
    assignments : vector of initial labels (-1 = unassigned) or one of
      "rand" (``np.random.randint(0, K, N)`` from the global stream),
      "one-by-one" (only X[0] seated), "each-in-own".
    K : initial number of components for "rand".  K_max : component slots.
    covariance_type : "full", "diag" or "fixed" (the latter with a ``FixedVarPrior``).
    device : GPU ordinal (extension).  rng / nprng : ``random.Random`` /
      ``np.random.RandomState`` to draw from instead of the process-global streams
      (extension, used for one-chain-per-GPU runs).
    """

    def __init__(self, X, kernel_prior, alpha, save_path, assignments="rand", K=1, K_max=None,
                 covariance_type="full", device=0, rng=None, nprng=None):
        super(IGMM, self).__init__()
        X = np.asarray(X)
        if len(X.shape) < 2:
            raise ValueError('X must be at least a 2-dimensional array.')
        self.save_path = save_path
        self.alpha = alpha
        self.N, self.D = X.shape
        self._rng = rng
        self._nprng = np.random if nprng is None else nprng
        self._lease = None              # the generators' states while a sampler loop runs (utils/rng.py: DeviceLease)

        if isinstance(assignments, str):
            if assignments == "rand":
                assignments = compact_labels(self._nprng.randint(0, K, self.N))
            elif assignments == "one-by-one":
                assignments = -1 * np.ones(self.N, dtype="int")
                assignments[0] = 0
            elif assignments == "each-in-own":
                assignments = np.arange(self.N)
            else:
                raise ValueError("unknown assignments mode %r" % (assignments,))

        if covariance_type == "full":
            self.components = GaussianComponents(X, kernel_prior, assignments, K_max,
                                                 device=device, alpha=alpha)
        elif covariance_type == "diag":
            self.components = GaussianComponentsDiag(X, kernel_prior, assignments, K_max,
                                                     device=device, alpha=alpha)
        elif covariance_type == "fixed":
            self.components = GaussianComponentsFixedVar(X, kernel_prior, assignments, K_max,
                                                         device=device, alpha=alpha)
        else:
            assert False, "Invalid covariance type."

    @classmethod
    def sample_chains(cls, X, kernel_prior, alpha, chains, n_iter, seed=0, device=0, true_assignments=None, **kwargs):
        """``chains`` independent chains of this model on ONE GPU (``chains=``), chain c seeded
        ``seed + c``, their sampler loops in lockstep so that a round of sweeps is one group call -- what pays wherever
        a sweep is one workgroup's chain of dependent draws (D <= 4 always; any dimension while the chains still move)
        and a GPU has 256 compute units.
        Returns ``[(model, record_dict), ...]``; see ``pybgmm_amd.chains.run_chains_on_device`` for the keywords."""
        from .. import chains as _chains
        return _chains.run_chains_on_device(cls, X, kernel_prior, alpha, chains, n_iter, seed=seed, device_index=device,
                                            true_assignments=true_assignments, **kwargs)

    # ------------------------------------------------------------------ #
    def setup_distribution_dict(self, num_saved):
        return {"mean": np.zeros(shape=(num_saved, 0)),
                "variance": np.zeros(shape=(num_saved, 0)),
                "weights": np.zeros(shape=(num_saved, 0))}

    def update_distribution_dict(self, distribution_dict, weight_first):
        """Snapshot of MAP means / variances / Dirichlet weights with the reference's
        label-switch ordering (igmm.py:128-197).  The matplotlib output of the
        reference for D == 2 is not produced; the ``np.random`` Dirichlet draw is,
        so the caller-visible stream stays aligned."""
        self._settle_generators()       # (the Dirichlet weights below come out of the caller's numpy stream)
        means, sds = [], []
        for mu, sigma in self.components.map_all():
            means.append(mu)
            sds.append(sigma)
        sds = np.array(sds).flatten()
        means = np.array(means).flatten()
        if weight_first:
            weights = self.gibbs_weight()
            idx = np.argsort(weights)
        else:
            idx = np.argsort(means)
            weights = self.gibbs_weight()
        means = self.label_switch(idx, means)
        sds = self.label_switch(idx, sds)
        weights = self.label_switch(idx, weights)
        self.old_mean, self.old_sigma = means, sds
        for key, val in (("mean", means), ("variance", sds), ("weights", weights)):
            distribution_dict[key] = np.hstack((distribution_dict[key], val.reshape((val.shape[0], 1))))
        return distribution_dict

    def log_marg(self):
        """log p(X, z) (igmm.py:199-215), evaluated on the device."""
        return self.components._ctx.log_marg()

    def log_marg_host(self):
        """The same quantity assembled on the host from downloaded counts and the
        per-component device marginals (cross-check for tests)."""
        K = self.components.K
        counts = self.components.counts[:K]
        facts_ = gammaln(counts)
        facts_[counts == 0] = 0
        log_prob_z = ((K - 1) * math.log(self.alpha) + gammaln(self.alpha)
                      - gammaln(np.sum(counts) + self.alpha) + np.sum(facts_))
        return log_prob_z + self.components.log_marg()

    def gibbs_weight(self):
        K = self.components.K
        Nk = self.components.counts[:K].tolist()
        alpha = [Nk[cid] + self.alpha / K for cid in range(K)]
        if self._nprng is np.random:
            return stats.dirichlet(alpha).rvs(size=1).flatten()
        return stats.dirichlet(alpha).rvs(size=1, random_state=self._nprng).flatten()

    # ------------------------------------------------------------------ #
    def _lease_generators(self):
        """Start of a sampler loop: from here to ``_settle_generators`` the device advances the generators' states."""
        self._lease = _rng.DeviceLease(self._rng, self._nprng)

    def _settle_generators(self):
        if self._lease is not None:
            self._lease.settle()

    def _release_generators(self):
        """End of a sampler loop (also on an exception): the generators are the caller's again."""
        if self._lease is not None:
            self._lease.settle()
            self._lease = None

    def _draw_order(self):
        """``np.random.permutation(range(N))`` from the caller's numpy stream as the next sweep's visiting order: drawn on
        the device where the library can (``rng.STAGED``), on the host otherwise (the array)."""
        if self._lease is not None:
            return self._lease.stage_permutation(self.components._ctx, self.N)
        return _rng.take_permutation_staged(self.components._ctx, self.N, self._nprng)

    def _sweep(self, order=None, power=None):
        """One device sweep fed from the caller's random streams."""
        ctx = self.components._ctx
        if order is _rng.STAGED:        # (drawn on the device and already in place: _draw_order)
            order = None
        if self._lease is not None:
            staged = self._lease.stage_uniforms(ctx, order)             # the caller's stream, continued on the GPU
        else:
            staged = _rng.stage_uniforms_on_device(ctx, order, self._rng)
        if not staged:
            ctx.stage(_rng.take_uniforms(self.N, self._rng), order)
        step = getattr(self, "_lockstep", None)
        try:
            if step is not None:        # one of several chains on this device (chains.run_chains_on_device)
                step.sweep(self, power)
            else:
                ctx.sweep_staged(power)
        except _lib.BGMMError as e:
            # ``K_max=None`` means "as many components as the chain opens, up to N" (reference
            # gaussian_components.py:81-83); the slots are grown on demand instead of set aside up front
            if e.code != -3 or not getattr(self.components, "K_max_auto", False) or self.components.K_max >= self.N:
                raise
            self._settle_generators()   # (the new context has no look-ahead of this stream)
            resume_in_larger_context(self.components, power)
