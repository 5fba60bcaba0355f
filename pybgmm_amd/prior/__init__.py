from .niw import NIW

__all__ = ["NIW"]
