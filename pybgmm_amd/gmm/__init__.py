from .gmm import GMM

__all__ = ["GMM"]
