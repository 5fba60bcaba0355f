"""
Collapsed Gibbs sampler for the powered Chinese restaurant process mixture model --
the interface of reference pybgmm/igmm/pcrpmm.py:20-192.
"""
import logging
import time

from ..utils import rng as _rng
from .igmm import IGMM

logger = logging.getLogger(__name__)


class PCRPMM(IGMM):

    def __init__(self, X, kernel_prior, alpha, save_path, assignments="rand", K=1, K_max=None,
                 covariance_type="full", **device_kwargs):
        super(PCRPMM, self).__init__(X, kernel_prior, alpha, save_path, assignments=assignments,
                                     K=K, K_max=K_max, covariance_type=covariance_type,
                                     **device_kwargs)

    def collapsed_gibbs_sampler(self, n_iter, true_assignments, n_power=1.01, power_burnin=0,
                                num_saved=3, weight_first=True, flag_power=True):
        """
        ``flag_power`` False makes this the plain CRPMM.  With it on (and ``n_power > 1``)
        every sweep -- burn-in sweeps included -- visits the data in a fresh
        ``np.random.permutation`` (pcrpmm.py:86-91), and sweeps with
        ``i_iter > power_burnin`` weigh an existing table by ``log(n_k ** n_power)``
        (pcrpmm.py:105-108); a new table keeps the un-powered ``log(alpha)``.
        """
        record_dict = self.setup_record_dict()
        start_time = time.time()
        distribution_dict = self.setup_distribution_dict(num_saved)
        self._lease_generators()
        try:
            for i_iter in range(n_iter):
                if num_saved == self.components.K and i_iter > 1:
                    distribution_dict = self.update_distribution_dict(distribution_dict, weight_first)
                if flag_power and n_power > 1:
                    if i_iter % 20 == 0:
                        logger.info(" Permutate data; " + "Power value: {}".format(n_power))
                    order = self._draw_order()
                else:
                    order = None
                power = n_power if (flag_power and i_iter > power_burnin) else None
                self._sweep(order=order, power=power)
                record_dict = self.update_record_dict(record_dict, i_iter, true_assignments, start_time)
                start_time = time.time()
        finally:
            self._release_generators()
        return record_dict, distribution_dict

    fit = collapsed_gibbs_sampler
