from .igmm import IGMM
from .crpmm import CRPMM
from .pcrpmm import PCRPMM

__all__ = ["IGMM", "CRPMM", "PCRPMM"]
