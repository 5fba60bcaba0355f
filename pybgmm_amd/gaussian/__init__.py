from .gaussian_components import GaussianComponents, GaussianComponentsDiag

__all__ = ["GaussianComponents", "GaussianComponentsDiag"]
