from .gaussian_components import GaussianComponents, GaussianComponentsDiag
from .gaussian_components_fixedvar import FixedVarPrior, GaussianComponentsFixedVar

__all__ = ["GaussianComponents", "GaussianComponentsDiag", "GaussianComponentsFixedVar", "FixedVarPrior"]
