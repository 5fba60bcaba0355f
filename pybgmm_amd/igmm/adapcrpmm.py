"""
Collapsed Gibbs sampler for the adaptive powered CRP mixture model -- the interface of
reference pybgmm/igmm/adapcrpmm.py:20-219.  Same device sweep as
PCRPMM; only the exponent is chosen per sweep on the host.
"""
import logging
import time

import numpy as np

from ..utils import rng as _rng
from .igmm import IGMM

logger = logging.getLogger(__name__)


class ADAPCRPMM(IGMM):

    def __init__(self, X, kernel_prior, alpha, save_path, assignments="rand", K=1, K_max=None,
                 covariance_type="full", **device_kwargs):
        super(ADAPCRPMM, self).__init__(X, kernel_prior, alpha, save_path, assignments=assignments,
                                        K=K, K_max=K_max, covariance_type=covariance_type,
                                        **device_kwargs)

    def collapsed_gibbs_sampler(self, n_iter, true_assignments, r_up=1.3, adapcrp_perct=0.04,
                                adapcrp_burnin=0, num_saved=3, weight_first=True, flag_adapcrp=True):
        """
        Before every sweep with ``i_iter > adapcrp_burnin`` the exponent becomes
        ``1 + (r_up - 1) * (share of components with at most N * adapcrp_perct points)``
        (adapcrpmm.py:100-104); the data are visited in a fresh permutation whenever that
        exponent exceeds 1 (:110-115) and the tables are weighted by ``log(n_k ** power)`` in the
        sweeps past the burn-in (:131-138).

        The reference reads the exponent before it has ever been assigned when
        ``flag_adapcrp`` is on and the first sweep is still inside the burn-in (its default
        ``adapcrp_burnin=0``), which ends that call with ``UnboundLocalError``; the same
        exception is raised here.
        """
        record_dict = self.setup_record_dict()
        start_time = time.time()
        distribution_dict = self.setup_distribution_dict(num_saved)
        adapcrp_power = None
        self._lease_generators()
        try:
            for i_iter in range(n_iter):
                if num_saved == self.components.K and i_iter > 1:
                    distribution_dict = self.update_distribution_dict(distribution_dict, weight_first)
                powered = flag_adapcrp and i_iter > adapcrp_burnin
                if powered:
                    # the exponent grows with the share of "small" clusters -- those holding at most
                    # adapcrp_perct of the data -- from 1 (none small) up to r_up (all small): adapcrpmm.py:100-103
                    sizes = np.asarray(self.components._ctx.counts())
                    share_small = np.count_nonzero(sizes <= adapcrp_perct * self.components.N) / float(sizes.size)
                    adapcrp_power = 1.0 + share_small * (r_up - 1.0)
                    if i_iter % 20 == 0:
                        logging.info('Ada-pCRP power: {}'.format(adapcrp_power))
                order = None
                if flag_adapcrp:
                    if adapcrp_power is None:
                        raise UnboundLocalError("local variable 'adapcrp_power' referenced before assignment")
                    if adapcrp_power > 1:
                        if i_iter % 20 == 0:
                            logger.info(" Permutate data")
                        order = self._draw_order()
                self._sweep(order=order, power=adapcrp_power if powered else None)
                record_dict = self.update_record_dict(record_dict, i_iter, true_assignments, start_time)
                start_time = time.time()
        finally:
            self._release_generators()
        return record_dict, distribution_dict

    fit = collapsed_gibbs_sampler
