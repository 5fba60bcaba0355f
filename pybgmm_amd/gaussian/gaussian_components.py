"""
Device-resident Gaussian components with an NIW prior (full covariance).

Same constructor, attributes and methods as the reference's
``pybgmm/gaussian/gaussian_components.py:22`` ``GaussianComponents``, but the
sufficient statistics live in HBM behind libbgmm_hip.so and every method is a
kernel call through the C-ABI.  Attribute reads download on demand.

Differences a caller can observe:
  * ``K_max=None`` means "up to N components" as in the reference (gaussian_components.py:81-83), but the N x D x D
    floats that default makes the reference allocate (:86-89) are not set aside up front: the device starts with
    ``max(1024, 4*K_init)`` slots (N for N <= 4096) and, when a sweep opens more components than that, the
    sampler loop moves the chain into a context with twice the slots and carries on at the very visit that needed
    the new slot (``resume_in_larger_context`` below) -- same trajectory, no ``BGMM_EKMAX`` short of N.
    An explicit ``K_max`` is a hard limit, as in the reference (there: IndexError).
  * ``_cached_outer`` (N x D x D, :116-118) is never materialised.
  * ``cache_component_stats`` / ``restore_component_from_stats`` download / upload one
    component's statistics (``bgmm_get_stats`` / ``bgmm_set_stats``); the sweep kernels do
    not need them -- a visit that stays is an exact no-op by construction.
"""
import numpy as np
from scipy.special import gammaln

from .. import _lib


def reference_tables(v_0, N):
    """The reference's ``_cached_gammaln_by_2`` / ``_cached_log_v`` tables
    (gaussian_components.py:120-122), computed with the same scipy / numpy calls."""
    n = np.concatenate([[1], np.arange(1, int(v_0) + N + 2)])
    return gammaln(n / 2.), np.log(n)


def default_K_max(N, K_init):
    """Slots a context starts with when the caller leaves ``K_max`` to the library (it grows on demand up to N)."""
    if N <= 4096:
        return N
    return int(min(N, max(1024, 4 * K_init)))


def resume_in_larger_context(comp, power):
    """A sweep of ``comp`` has just failed with BGMM_EKMAX and ``K_max`` was left to the library: continue it.

    What the failed sweep leaves (tests: ``test_k_max_overflow_leaves_a_consistent_state``): every visit in front of the
    failing one applied, the failing visit's point taken out of its component (-1, as after the reference's ``del_item``,
    gaussian_components.py:171-186) and nothing else; its inputs are still staged.  So: a context with twice the slots,
    the labels and the raw statistics put back bit for bit (the checkpoint / resume route of SURVEY.md section 5), the
    staged uniforms and visiting order rotated so that the failing visit comes first, and a partial sweep over the
    visits that were left (``bgmm_set_sweep_visits``).  Repeats if the rest of the sweep outgrows the new context too.
    Returns nothing; ``comp._ctx`` is the new context afterwards and ``comp.K_max`` its slot count."""
    n_left = comp.N                                    # visits of the sweep that failed
    while True:
        old = comp._ctx
        if comp.K_max >= comp.N:
            raise _lib.BGMMError(-3, "K_max exceeded with K_max == N")
        u = old.staged_uniforms()
        try:
            order = old.staged_order()
        except _lib.BGMMError:
            order = np.arange(comp.N, dtype=np.int64)
        z = old.assignments()
        # the failing visit: the first one in visiting order whose point is unassigned (the points a "one-by-one" start
        # has not reached yet are unassigned too -- they come later)
        unassigned = np.nonzero(z[order[:n_left]] < 0)[0]
        assert unassigned.size >= 1, "BGMM_EKMAX without an unassigned point"
        p = int(unassigned[0])
        K = old.K
        counts = old.counts()
        raw = [old.raw_stats(k) for k in range(K)]
        new_K_max = int(min(comp.N, max(2 * comp.K_max, K + 1)))
        new = comp._new_context(new_K_max)
        new.set_assignments(z)
        for k in range(K):
            new.set_stats(k, raw[k][0], raw[k][1], int(counts[k]))
        old.close()
        comp._ctx, comp.K_max = new, new_K_max
        n_left -= p
        new.stage(np.roll(u, -p), np.roll(order, -p))
        new.set_sweep_visits(n_left)
        try:
            new.sweep_staged(power)
            return
        except _lib.BGMMError as e:
            if e.code != -3:
                raise


class GaussianComponents(object):
    """``alpha`` (extension): CRP concentration used by the sweep kernel for the
    "open a new table" score; the reference keeps it on the sampler object."""

    _cov_type = "full"

    def __init__(self, X, prior, assignments=None, K_max=None, device=0, alpha=1.0):
        X = np.asarray(X)
        self.X = X
        self.prior = prior
        self.N, self.D = X.shape
        if assignments is None:
            z = -1 * np.ones(self.N, dtype=np.int64)
        else:
            z = np.asarray(assignments, dtype=np.int64)
            assert (self.N,) == z.shape
            # apart from unassigned (-1), components should be labelled from 0
            assert set(z.tolist()).difference([-1]) == set(range(int(z.max()) + 1))
        K_init = int(z.max()) + 1
        self.K_max_auto = K_max is None             # (the library picks the slots and grows them on demand, up to N)
        if K_max is None:
            K_max = default_K_max(self.N, K_init)
        self.K_max = int(K_max)
        assert K_init <= self.K_max, "initial assignments use more than K_max components"
        self._cached_log_pi = np.log(np.pi)
        self._cached_gammaln_by_2, self._cached_log_v = reference_tables(prior.v_0, self.N)
        self._check_prior(prior)
        self._device, self._alpha = device, alpha
        self._ctx = self._new_context(self.K_max)
        self._log_prior = None
        self._ctx.set_assignments(z)

    def _new_context(self, K_max):
        return _lib.Context(self.X, self.prior.m_0, self.prior.k_0, self.prior.v_0, self.prior.S_0,
                            self._alpha, K_max, device=self._device,
                            tables=(self._cached_gammaln_by_2, self._cached_log_v),
                            cov_type=self._cov_type)

    def _check_prior(self, prior):
        pass

    def _block(self):
        return (self.D, self.D)

    # --- attributes of the reference class, served from the device ------------------
    @property
    def K(self):
        return self._ctx.K

    @property
    def assignments(self):
        return self._ctx.assignments()

    @property
    def counts(self):
        out = np.zeros(self.K_max, dtype=np.int64)
        c = self._ctx.counts()
        out[:len(c)] = c
        return out

    def _padded(self, arr, shape):
        out = np.zeros((self.K_max,) + shape)
        out[:arr.shape[0]] = arr
        return out

    @property
    def m_N_numerators(self):
        return self._padded(self._ctx.stats(False)[0], (self.D,))

    @property
    def S_N_partials(self):
        return self._padded(self._ctx.stats(False)[1], self._block())

    @property
    def logdet_covars(self):
        return self._padded(self._ctx.stats(False)[2], ())

    @property
    def inv_covars(self):
        return self._padded(self._ctx.stats(True)[3], self._block())

    @property
    def cached_log_prior(self):
        if self._log_prior is None:
            self._log_prior = self._ctx.log_prior()
        return self._log_prior

    # --- methods ------------------------------------------------------------------
    def add_item(self, i, k):
        """Add data vector ``X[i]`` to component ``k`` (``k == K`` opens a new one)."""
        self._ctx.add_item(i, k)

    def del_item(self, i):
        """Remove data vector ``X[i]`` from its component."""
        self._ctx.del_item(i)

    def del_component(self, k):
        """Remove component ``k`` (gaussian_components.py:188-205: the last label takes its place).  The reference only
        calls this on a component that has just lost its last member; members it still has become unassigned (-1)
        here -- the reference would leave them labelled ``k``, i.e. in the component that moved in (include/bgmm.h)."""
        self._ctx.del_component(k)

    def cache_component_stats(self, k):
        m, S, ld, iv = self._ctx.stats(True)
        return (m[k].copy(), S[k].copy(), ld[k], iv[k].copy(), int(self._ctx.counts()[k]))

    def restore_component_from_stats(self, k, m_N_numerator, S_N_partial, logdet_covar, inv_covar, count):
        """Restore component ``k`` from statistics taken with ``cache_component_stats``
        (gaussian_components.py:144-152).  The statistics and the count are written to the GPU as
        given; ``logdet_covar`` / ``inv_covar`` are rebuilt from them there (``bgmm_set_stats``)."""
        self._ctx.set_stats(k, m_N_numerator, S_N_partial, count)

    def set_assignment(self, i, k):
        """``components.assignments[i] = k`` of the reference's sampler loop (crpmm.py:85): the
        ``assignments`` attribute here is a download, so the write has its own method."""
        self._ctx.set_label(i, k)

    def log_prior(self, i):
        """Probability of ``X[i]`` under the prior alone."""
        return float(self.cached_log_prior[i])

    def log_post_pred(self, i):
        """K-vector of the posterior predictive of ``X[i]`` under all components."""
        return self._ctx.log_post_pred(i)

    def log_post_pred_k(self, i, k):
        return float(self._ctx.log_post_pred(i)[k])

    def log_marg_k(self, k):
        return self._ctx.log_marg_k(k)

    def log_marg(self):
        """log p(X | z): sum of the per-component marginals."""
        return float(sum(self._ctx.log_marg_k(k) for k in range(self.K)))

    def map_all(self):
        """``[map(k) for k in range(K)]`` from ONE download of the statistics (a distribution-dict
        snapshot asks for every component)."""
        m, S, _, _ = self._ctx.stats(False)
        counts = self._ctx.counts()
        out = []
        for k in range(len(counts)):
            k_N = self.prior.k_0 + int(counts[k])
            v_N = self.prior.v_0 + int(counts[k])
            m_N = m[k] / k_N
            out.append((m_N, (S[k] - k_N * np.outer(m_N, m_N)) / (v_N + self.D + 2)))
        return out

    def map(self, k):
        """MAP estimate (mean, covariance) of component ``k`` (Murphy 4.215)."""
        m, S, _, _ = self._ctx.stats(False)
        n = int(self._ctx.counts()[k])
        k_N = self.prior.k_0 + n
        v_N = self.prior.v_0 + n
        m_N = m[k] / k_N
        sigma = (S[k] - k_N * np.outer(m_N, m_N)) / (v_N + self.D + 2)
        return (m_N, sigma)

    def rand_k(self, k, rng=None, nprng=None):
        """A random (mean, covariance) from the posterior NIW of component ``k``
        (gaussian_components.py:291-303): consumes the caller's ``np.random`` / ``random`` streams
        exactly as the reference does (prior/wishart.py)."""
        from ..prior import wishart
        nprng_ = np.random if nprng is None else nprng
        m, S, _, _ = self._ctx.stats(False)
        n = int(self._ctx.counts()[k])
        k_N = self.prior.k_0 + n
        v_N = self.prior.v_0 + n
        m_N = m[k] / k_N
        S_N = S[k] - k_N * np.outer(m_N, m_N)
        sigma = np.linalg.solve(np.linalg.cholesky(S_N).T, np.eye(self.D))
        sigma = wishart.iwishrnd(sigma, v_N, sigma, rng=rng, nprng=nprng)
        mu = nprng_.multivariate_normal(m_N, sigma / k_N)
        return mu, sigma


def log_post_pred_unvectorized(gmm, i):
    """
    The cross-check helper of the reference (gaussian_components.py:355-363: ``log_post_pred`` one component at a time,
    "for testing purposes").  ``log_post_pred`` / ``log_post_pred_k`` here are one kernel's output, so a per-component
    loop over them would check nothing; this one is independent of the device's predictive: it takes the raw
    statistics (counts, ``m``, ``S``) from the GPU and evaluates every component's Student-t on the HOST the way the
    reference's scalar path does -- covariance ``(k_N + 1) / (k_N (v_N - D + 1)) (S - k_N m_N m_N')`` (:319-331),
    its ``slogdet`` and inverse by LAPACK, the density of ``_multivariate_students_t`` (:334-344).
    Full-covariance components only (the class the reference defines it for).
    """
    assert type(gmm) is GaussianComponents, "defined for full-covariance components, as in the reference"
    D, prior = gmm.D, gmm.prior
    counts = gmm._ctx.counts()
    m, S, _, _ = gmm._ctx.stats(False)
    x = np.asarray(gmm.X[i], dtype=np.float64)
    out = np.zeros(len(counts), dtype=np.float64)
    for k in range(len(counts)):
        n = int(counts[k])
        k_N, v_N = prior.k_0 + n, prior.v_0 + n
        mean = m[k] / k_N
        dof = v_N - D + 1
        covar = (k_N + 1.0) / (k_N * dof) * (S[k] - k_N * np.outer(mean, mean))
        diff = x - mean
        out[k] = (gammaln((dof + D) / 2.0) - gammaln(dof / 2.0) - D / 2.0 * np.log(dof) - D / 2.0 * np.log(np.pi)
                  - 0.5 * np.linalg.slogdet(covar)[1]
                  - (dof + D) / 2.0 * np.log1p(diff.dot(np.linalg.inv(covar)).dot(diff) / dof))
    return out


class GaussianComponentsDiag(GaussianComponents):
    """
    Diagonal-covariance components: the interface of the reference's
    ``pybgmm/gaussian/gaussian_components_diag.py:19`` ``GaussianComponentsDiag``.  ``prior.S_0``
    is a D-vector; ``S_N_partials`` / ``inv_vars`` are K_max x D, ``log_prod_vars`` K_max.
    The predictive is a product of univariate Student-t densities with ``v_N`` degrees of freedom.
    """
    _cov_type = "diag"

    def _check_prior(self, prior):
        assert len(np.asarray(prior.S_0).shape) == 1, "For diagonal covariance, S_0 needs to be vector."

    def _block(self):
        return (self.D,)

    @property
    def log_prod_vars(self):
        return self._padded(self._ctx.stats(False)[2], ())

    @property
    def inv_vars(self):
        return self._padded(self._ctx.stats(True)[3], (self.D,))

    # the full-covariance names do not exist on the reference's diag class
    logdet_covars = property(lambda self: (_ for _ in ()).throw(AttributeError("logdet_covars")))
    inv_covars = property(lambda self: (_ for _ in ()).throw(AttributeError("inv_covars")))

    def map(self, k):
        raise NotImplementedError("the reference's diagonal class has no map()")

    def map_all(self):
        raise NotImplementedError("the reference's diagonal class has no map()")

    def rand_k(self, k, rng=None, nprng=None):
        """A random (mean vector, variance vector) from the posterior product of
        normal-inverse-chi-squared distributions of component ``k``
        (gaussian_components_diag.py:305-322, 388-400): per dimension one ``np.random.gamma`` and one
        ``np.random.normal``, in that order.  The reference is Python 2: its ``alpha = df/2`` is an
        integer division for the integer degrees of freedom it is called with, kept here."""
        nprng = np.random if nprng is None else nprng
        m, S, _, _ = self._ctx.stats(False)
        n = int(self._ctx.counts()[k])
        k_N = self.prior.k_0 + n
        v_N = self.prior.v_0 + n
        m_N = m[k] / k_N
        S_N = S[k] - k_N * np.square(m_N)
        mean, var = np.zeros(self.D), np.zeros(self.D)
        for i in range(self.D):
            scale = S_N[i] / v_N
            # (Python 2: only an int-typed v_N floor-divides; a float v_0 = 5.0 gives true division)
            a = v_N // 2 if isinstance(v_N, (int, np.integer)) else v_N / 2.0
            var[i] = 1.0 / nprng.gamma(a, 1.0 / (v_N * scale / 2.0), 1)[0]
            mean[i] = nprng.normal(m_N[i], np.sqrt(var[i] / k_N))
        return mean, var
