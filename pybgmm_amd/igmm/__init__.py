from .igmm import IGMM
from .crpmm import CRPMM
from .pcrpmm import PCRPMM
from .adapcrpmm import ADAPCRPMM

__all__ = ["IGMM", "CRPMM", "PCRPMM", "ADAPCRPMM"]
