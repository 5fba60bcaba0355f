"""
Fixed-variance Gaussian components: the interface of the reference's
``pybgmm/gaussian/gaussian_components_fixedvar.py:18`` ``GaussianComponentsFixedVar`` and
``FixedVarPrior`` (:304-311).  Mean-only conjugate update with a known per-dimension variance;
the predictive is a product of univariate normals.  Statistics live on the GPU.
"""
import numpy as np

from .. import _lib
from .gaussian_components import GaussianComponents, default_K_max


class FixedVarPrior(object):
    """The prior parameters for a fixed diagonal covariance multivariate Gaussian."""

    def __init__(self, var, mu_0, var_0):
        self.var = var
        self.mu_0 = mu_0
        self.var_0 = var_0


class GaussianComponentsFixedVar(GaussianComponents):
    _cov_type = "fixed"

    def __init__(self, X, prior, assignments=None, K_max=None, device=0, alpha=1.0):
        X = np.asarray(X)
        self.X = X
        self.prior = prior
        self.N, self.D = X.shape
        self.precision = 1. / np.asarray(prior.var, dtype=np.float64)
        self.mu_0 = np.asarray(prior.mu_0, dtype=np.float64)
        self.precision_0 = 1. / np.asarray(prior.var_0, dtype=np.float64)
        if assignments is None:
            z = -1 * np.ones(self.N, dtype=np.int64)
        else:
            z = np.asarray(assignments, dtype=np.int64)
            assert (self.N,) == z.shape
            assert set(z.tolist()).difference([-1]) == set(range(int(z.max()) + 1))
        K_init = int(z.max()) + 1
        self.K_max_auto = K_max is None
        if K_max is None:
            K_max = default_K_max(self.N, K_init)
        self.K_max = int(K_max)
        assert K_init <= self.K_max, "initial assignments use more than K_max components"
        self._device, self._alpha = device, alpha
        self._ctx = self._new_context(self.K_max)
        self._log_prior = None
        self._ctx.set_assignments(z)

    def _new_context(self, K_max):
        S_0 = np.concatenate([np.broadcast_to(np.asarray(self.prior.var, dtype=np.float64), (self.D,)),
                              np.broadcast_to(np.asarray(self.prior.var_0, dtype=np.float64), (self.D,))])
        return _lib.Context(self.X, np.broadcast_to(self.mu_0, (self.D,)), 1.0, 1, S_0, self._alpha, K_max,
                            device=self._device, cov_type="fixed")

    def _block(self):
        return (self.D,)

    @property
    def mu_N_numerators(self):
        return self._padded(self._ctx.stats(False)[0], (self.D,))

    @property
    def precision_Ns(self):
        return self._padded(self._ctx.stats(False)[1], (self.D,))

    @property
    def log_prod_precision_preds(self):
        return self._padded(self._ctx.stats(False)[2], ())

    @property
    def precision_preds(self):
        return self._padded(self._ctx.stats(True)[3], (self.D,))

    def _no(name):
        return property(lambda self: (_ for _ in ()).throw(AttributeError(name)))

    m_N_numerators = _no("m_N_numerators")
    S_N_partials = _no("S_N_partials")
    logdet_covars = _no("logdet_covars")
    inv_covars = _no("inv_covars")
    del _no

    def cache_component_stats(self, k):
        """The reference's five statistics of component ``k`` (gaussian_components_fixedvar.py:124-134).  The library also
        keeps the sum of squares of the members (its log marginal needs it; the reference recomputes from X): it is
        remembered here, so that ``restore_component_from_stats`` puts back exactly what was there."""
        m, pN, lpp, pp = self._ctx.stats(True)
        raw_m, raw_S = self._ctx.raw_stats(k)
        if not hasattr(self, "_raw_cache"):
            self._raw_cache = {}
        self._raw_cache[int(k)] = (raw_m, raw_S)
        return (m[k].copy(), pN[k].copy(), lpp[k], pp[k].copy(), int(self._ctx.counts()[k]))

    def map(self, k):
        raise NotImplementedError("the reference's fixed-variance class has no map()")

    def map_all(self):
        raise NotImplementedError("the reference's fixed-variance class has no map()")

    def restore_component_from_stats(self, k, mu_N_numerator, precision_N, log_prod_precision_pred, precision_pred, count):
        """Restore component ``k`` (gaussian_components_fixedvar.py:136-144): numerators, precisions and the count are
        written to the GPU as given, the predictive's constants rebuilt there.  The members' sum of squares comes from
        the matching ``cache_component_stats`` call when there was one (the cache / del_item / restore idiom of the
        sampler loop), else from the points currently labelled ``k``."""
        mu = np.ascontiguousarray(mu_N_numerator, dtype=np.float64)
        pN = np.ascontiguousarray(precision_N, dtype=np.float64)
        cached = getattr(self, "_raw_cache", {}).get(int(k))
        if cached is not None and np.array_equal(cached[0], mu) and np.array_equal(cached[1][:self.D], pN):
            sumsq = cached[1][self.D:]
        else:
            members = self._ctx.assignments() == k
            sumsq = np.square(self.X[members]).sum(axis=0) if members.any() else np.zeros(self.D)
        self._ctx.set_stats(k, mu, np.concatenate([pN, sumsq]), count)

    def rand_k(self, k, rng=None, nprng=None):
        """A random mean vector from the posterior product of normals of component ``k``
        (gaussian_components_fixedvar.py:266-276): D draws of ``np.random.normal``."""
        nprng = np.random if nprng is None else nprng
        m, pN, _, _ = self._ctx.stats(False)
        mu_N = m[k] / pN[k]
        var_N = 1.0 / pN[k]
        mean = np.zeros(self.D)
        for i in range(self.D):
            mean[i] = nprng.normal(mu_N[i], np.sqrt(var_N[i]))
        return mean
