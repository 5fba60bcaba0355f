"""
``pybgmm`` -- the reference's import path, served by ``pybgmm_amd``.

    from pybgmm.prior import NIW
    from pybgmm.igmm import CRPMM, PCRPMM

run unchanged (reference pybgmm/igmm/__init__.py:7-8, pybgmm/prior/__init__.py:2): every
``pybgmm.<sub>`` module IS the ``pybgmm_amd.<sub>`` module -- the same object under a second name in
``sys.modules``, not a re-export file -- so classes compare identical whichever path imported them.
Only the sub-packages on the collapsed-Gibbs path exist (igmm, prior, gaussian, gmm, utils); the
reference's plotting, ARS and the CSCRPMM / SubCRPMM samplers are out of scope and
importing them raises ImportError as for any missing module.  There is no CPU fallback behind this
name either: constructing a sampler without a HIP device raises.
"""
import importlib
import sys

import pybgmm_amd as _impl

__version__ = _impl.__version__

_ALIASED = ("igmm", "prior", "gaussian", "gmm", "utils",
            "igmm.igmm", "igmm.crpmm", "igmm.pcrpmm", "igmm.adapcrpmm",
            "prior.niw", "prior.wishart",
            "gaussian.gaussian_components", "gaussian.gaussian_components_fixedvar",
            "gmm.gmm", "utils.gendata", "utils.metrics")

for _name in _ALIASED:
    _mod = importlib.import_module("pybgmm_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        globals()[_name] = _mod

# the reference's `from pybgmm.utils import gendata_1d` (pybgmm/utils/__init__.py:1)
if not hasattr(sys.modules[__name__ + ".utils"], "gendata_1d"):
    sys.modules[__name__ + ".utils"].gendata_1d = sys.modules[__name__ + ".utils.gendata"].gendata_1d

del _name, _mod
