"""
Collapsed Gibbs sampler for the Chinese restaurant process mixture model --
the interface of reference pybgmm/igmm/crpmm.py:15-94, one kernel-driven sweep
per iteration instead of a Python loop over datapoints.
"""
import time

from .igmm import IGMM


class CRPMM(IGMM):

    def __init__(self, X, kernel_prior, alpha, save_path, assignments="rand", K=1, K_max=None,
                 covariance_type="full", **device_kwargs):
        super(CRPMM, self).__init__(X, kernel_prior, alpha, save_path, assignments=assignments,
                                    K=K, K_max=K_max, covariance_type=covariance_type,
                                    **device_kwargs)

    def collapsed_gibbs_sampler(self, n_iter, true_assignments, num_saved=3, weight_first=True):
        """
        Perform ``n_iter`` sweeps.  Returns ``(record_dict, distribution_dict)`` with the
        reference's keys; the distribution dict is extended whenever the number of
        components equals ``num_saved`` (after the second sweep), as in crpmm.py:49-50.
        Every sweep visits the datapoints in index order and consumes exactly N
        ``random.random()`` values from the caller's stream.
        """
        record_dict = self.setup_record_dict()
        start_time = time.time()
        distribution_dict = self.setup_distribution_dict(num_saved)
        self._lease_generators()
        try:
            for i_iter in range(n_iter):
                if num_saved == self.components.K and i_iter > 1:
                    distribution_dict = self.update_distribution_dict(distribution_dict, weight_first)
                self._sweep(order=None, power=None)
                record_dict = self.update_record_dict(record_dict, i_iter, true_assignments, start_time)
                start_time = time.time()
        finally:
            self._release_generators()
        return record_dict, distribution_dict

    fit = collapsed_gibbs_sampler
