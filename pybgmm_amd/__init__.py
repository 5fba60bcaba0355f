"""
pybgmm_amd -- MI355X-native collapsed Gibbs sampling for CRP / pCRP Gaussian
mixtures behind the PyBGMM class surface (``NIW``, ``CRPMM``, ``PCRPMM``).

    from pybgmm_amd.prior import NIW
    from pybgmm_amd.igmm import CRPMM, PCRPMM
"""
__version__ = "0.1"
