from .gaussian_components import GaussianComponents

__all__ = ["GaussianComponents"]
