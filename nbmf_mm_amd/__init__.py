"""nbmf_mm_amd — MI355X-native NBMF-MM (drop-in for ``nbmf_mm``'s NumPy path).

Export list mirrors src/nbmf_mm/__init__.py:10-17 of the reference.
"""
from ._base import NBMF, NBMFMM
from ._solver import nbmf_mm_solver

__version__ = "0.1.0"
__all__ = ["NBMFMM", "NBMF", "nbmf_mm_solver"]
